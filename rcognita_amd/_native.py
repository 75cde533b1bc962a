"""ctypes binding of librcg.so (C ABI: include/rcg.h).

The library is built in-tree (``make lib`` or ``__graft_entry__.build()``) at
``rcognita_amd/lib/librcg.so``.  There is no fallback of any kind: if the shared object is missing,
or no HIP device is visible when a handle is created, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# The in-tree library.  The binding reads NO environment variable; a tool that wants the -DRCG_DEV twin (librcg_dev.so, the
# launcher's A/B knobs) says so in its own code, before the first handle: use_library(path).
LIB_PATH = os.path.join(_HERE, "lib", "librcg.so")

# ---- enums (include/rcg.h) -------------------------------------------------------------------
RCG_VERSION = 119
OK, ERR_BAD_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_NONFINITE = 0, -1, -2, -3, -4, -5
SYS_3WROBOT, SYS_3WROBOT_NI, SYS_2TANK = 0, 1, 2
MODE_MPC, MODE_RQL, MODE_SQL = 0, 1, 2
STAGE_QUADRATIC, STAGE_BIQUADRATIC = 0, 1
CRITIC_QUAD_LIN, CRITIC_QUADRATIC, CRITIC_QUAD_NOMIX, CRITIC_QUAD_MIX = 0, 1, 2, 3
F32, F64 = 0, 1
HOST, DEVICE = 0, 1
FLAG_HAS_TARGET, FLAG_PER_ENV_PARS, FLAG_REF_LAG, FLAG_ACCUM_EVERY_SUBSTEP, FLAG_NO_CLIP = 1, 2, 4, 8, 16
FLAG_DISTURB = 32
DIM_DISTURB = {0: 2, 1: 2, 2: 1}  # sys_id -> dim_disturb (presets/main_*.py)

(FIELD_STATE, FIELD_ACTION, FIELD_ACCUM, FIELD_STEP_IDX, FIELD_EPISODE_IDX, FIELD_STATUS, FIELD_PARS,
 FIELD_STATE_INIT, FIELD_STATE_PREV, FIELD_BEST_J, FIELD_BEST_IDX, FIELD_W_CRITIC, FIELD_W_PREV, FIELD_OBS_BUF,
 FIELD_ACT_BUF, FIELD_RETURNS, FIELD_ACTION_SQN, FIELD_DISTURB, FIELD_SUBSTEP_IDX) = range(19)
FIELD_COUNT = 19

MODE_IDS = {"MPC": MODE_MPC, "RQL": MODE_RQL, "SQL": MODE_SQL}
STAGE_IDS = {"quadratic": STAGE_QUADRATIC, "biquadratic": STAGE_BIQUADRATIC}
CRITIC_IDS = {"quad-lin": CRITIC_QUAD_LIN, "quadratic": CRITIC_QUADRATIC, "quad-nomix": CRITIC_QUAD_NOMIX,
              "quad-mix": CRITIC_QUAD_MIX}
SYS_IDS = {"3wrobot": SYS_3WROBOT, "3wrobotNI": SYS_3WROBOT_NI, "2tank": SYS_2TANK}
SYS_DIMS = {SYS_3WROBOT: (5, 2, 2), SYS_3WROBOT_NI: (3, 2, 0), SYS_2TANK: (2, 1, 5)}  # ds, du, n_pars

# every symbol include/rcg.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "rcg_version", "rcg_last_error", "rcg_device_count", "rcg_create", "rcg_destroy", "rcg_set_stream", "rcg_use_own_stream",
    "rcg_synchronize", "rcg_dev_alloc", "rcg_dev_free", "rcg_memcpy_h2d", "rcg_memcpy_d2h", "rcg_set_field",
    "rcg_get_field", "rcg_field_bytes", "rcg_field_ptr", "rcg_rhs", "rcg_stage_obj", "rcg_critic",
    "rcg_actor_cost", "rcg_critic_cost", "rcg_sim_step", "rcg_sim_step_h", "rcg_actor_argmin", "rcg_control_tick",
    "rcg_critic_update", "rcg_control_ticks", "rcg_control_tick_n", "rcg_actor_optimize", "rcg_control_tick_opt", "rcg_nominal_action",
    "rcg_control_tick_nominal", "rcg_rhs_full", "rcg_disturb_noise", "rcg_episode_reset", "rcg_episode_stats", "rcg_tick_count", "rcg_set_tick_count", "rcg_profile", "rcg_profile_read",
    "rcg_profile_samples", "rcg_last_launch", "rcg_kernel_name", "rcg_wait_stream", "rcg_nominal_theta", "rcg_set_optimizer", "rcg_set_optimizer_tol", "rcg_set_tick_parts", "rcg_join", "rcg_loop_step", "rcg_loop_step_begin", "rcg_loop_step_end",
    "rcg_actor_search", "rcg_control_tick_search", "rcg_candidates_sample", "rcg_release_stream",
]
KERNEL_ACTOR, KERNEL_SIM, KERNEL_CRITIC = 0, 1, 2
# rcg_kernel_id (rcg_last_launch)
(KID_NONE, KID_ACTOR, KID_ACTOR_DMA, KID_TICKS, KID_ACTOR_OPT, KID_NOMINAL, KID_SIM, KID_SIM_V, KID_SIM_DIST,
 KID_CRITIC_FIT, KID_ACTOR_DMA_PACKED, KID_ACTOR_SEARCH) = range(12)
DMA_MPC_G1, DMA_MPC, DMA_RQL_0, DMA_SQL_0 = 0, 1, 2, 6  # variant of k_actor_dma (+ critic_struct; rcg_actor_dma.hpp)
LOOP_DECIDE, LOOP_PUSH, LOOP_FIT = 1, 2, 4  # rcg_loop_step flags
DMA_RQL_GEN_0 = 12  # + critic_struct: RQL with a stage cost no preset has
DMA_MPC_GEND, DMA_MPC_GENF = 10, 11  # MPC with a cost structure no preset has: diagonal (biquadratic / target) | full matrices


class RcgCfg(C.Structure):
    """``rcg_cfg`` of include/rcg.h (field order and types must match exactly)."""

    _fields_ = [
        ("struct_size", C.c_int32), ("sys_id", C.c_int32), ("batch", C.c_int32), ("dtype", C.c_int32),
        ("device", C.c_int32), ("n_actor", C.c_int32), ("mode", C.c_int32), ("stage_obj_struct", C.c_int32),
        ("critic_struct", C.c_int32), ("n_critic", C.c_int32), ("buffer_size", C.c_int32),
        ("substeps_per_tick", C.c_int32), ("flags", C.c_int32), ("critic_every_ticks", C.c_int32),
        ("dt_sim", C.c_double), ("sampling_time", C.c_double), ("pred_step_size", C.c_double), ("gamma", C.c_double),
        ("pars", C.c_double * 8), ("ctrl_bnds", C.c_double * 4), ("R1", C.c_double * 49), ("R2", C.c_double * 49),
        ("target", C.c_double * 8), ("action_init", C.c_double * 4), ("w_init", C.c_double * 40),
        ("w_min", C.c_double * 40), ("w_max", C.c_double * 40),
        ("pars_disturb", C.c_double * 6), ("disturb_init", C.c_double * 2), ("seed", C.c_uint64),
        ("env_id_base", C.c_int64),
    ]


class RcgSummary(C.Structure):
    _fields_ = [("count", C.c_double), ("sum", C.c_double), ("sumsq", C.c_double), ("min", C.c_double),
                ("max", C.c_double), ("n_failed", C.c_double)]


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librcg error {code}: {msg}")
        self.code = code


_lib = None


def use_library(path):
    """Bind another build of the same ABI (tools/ and tests/knob_probe.py: librcg_dev.so) instead of the in-tree
    production library.  Must be called before anything has loaded the library."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_library: librcg is already loaded")
    LIB_PATH = os.path.abspath(path)


def lib():
    """Load librcg.so once.  Raises if it has not been built - there is no Python fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first (`make lib` at the repo root, or "
            "`python -c 'import __graft_entry__ as g; g.build()'`).  rcognita_amd has no CPU fallback."
        )
    # PyTorch-ROCm ships its own HIP runtime.  If this library brings the system runtime up first, torch's later
    # initialisation in the same process finds no GPU ("No HIP GPUs are available"); the other order works.  So when
    # torch is already imported, let it initialise first.  (torch is plumbing for callers that hand over tensors;
    # this module never imports it on its own.)
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and not torch.cuda.is_initialized():
        torch.cuda.init()
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    sig = {
        "rcg_version": (C.c_int, []),
        "rcg_last_error": (C.c_char_p, [vp]),
        "rcg_device_count": (C.c_int, []),
        "rcg_create": (C.c_int, [C.POINTER(RcgCfg), C.POINTER(vp)]),
        "rcg_destroy": (C.c_int, [vp]),
        "rcg_set_stream": (C.c_int, [vp, vp]),
        "rcg_use_own_stream": (C.c_int, [vp]),
        "rcg_synchronize": (C.c_int, [vp]),
        "rcg_dev_alloc": (C.c_int, [vp, u64, C.POINTER(vp)]),
        "rcg_dev_free": (C.c_int, [vp, vp]),
        "rcg_memcpy_h2d": (C.c_int, [vp, vp, vp, u64]),
        "rcg_memcpy_d2h": (C.c_int, [vp, vp, vp, u64]),
        "rcg_set_field": (C.c_int, [vp, C.c_int, vp, C.c_int]),
        "rcg_get_field": (C.c_int, [vp, C.c_int, vp, C.c_int]),
        "rcg_field_bytes": (i64, [vp, C.c_int]),
        "rcg_field_ptr": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
        "rcg_rhs": (C.c_int, [vp, vp, vp, vp, vp, i32, i32]),
        "rcg_stage_obj": (C.c_int, [vp, vp, vp, vp, i32]),
        "rcg_critic": (C.c_int, [vp, vp, vp, vp, vp, i32]),
        "rcg_actor_cost": (C.c_int, [vp, vp, i32, vp, vp, vp, vp]),
        "rcg_critic_cost": (C.c_int, [vp, vp, vp]),
        "rcg_sim_step": (C.c_int, [vp, i32]),
        "rcg_sim_step_h": (C.c_int, [vp, i32, C.c_double]),
        "rcg_actor_argmin": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, vp]),
        "rcg_control_tick": (C.c_int, [vp, vp, i32]),
        "rcg_critic_update": (C.c_int, [vp, i32]),
        "rcg_control_ticks": (C.c_int, [vp, i32, i32]),
        "rcg_control_tick_n": (C.c_int, [vp, vp, i32, i32]),
        "rcg_actor_optimize": (C.c_int, [vp, i32, vp, vp, vp, vp, vp, vp, vp]),
        "rcg_control_tick_opt": (C.c_int, [vp, i32, i32]),
        "rcg_set_optimizer": (C.c_int, [vp, i32]),
        "rcg_set_tick_parts": (C.c_int, [vp, i32]),
        "rcg_set_optimizer_tol": (C.c_int, [vp, C.c_double]),
        "rcg_loop_step_begin": (C.c_int, [vp, vp, C.c_double, i32, i32, i32]),
        "rcg_loop_step_end": (C.c_int, [vp, vp]),
        "rcg_loop_step": (C.c_int, [vp, vp, C.c_double, i32, i32, i32, vp]),  # (host double pointers passed as addresses)
        "rcg_join": (C.c_int, [vp]),
        "rcg_actor_search": (C.c_int, [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]),
        "rcg_control_tick_search": (C.c_int, [vp, i32, i32, i32]),
        "rcg_candidates_sample": (C.c_int, [vp, vp, i32, i32, vp]),
        "rcg_nominal_action": (C.c_int, [vp, vp, vp, vp, i32, C.c_double, C.POINTER(C.c_double), i32]),
        "rcg_control_tick_nominal": (C.c_int, [vp, C.c_double, C.POINTER(C.c_double)]),
        "rcg_nominal_theta": (C.c_int, [vp, vp, vp, i32]),
        "rcg_rhs_full": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32]),
        "rcg_disturb_noise": (C.c_int, [vp, vp, vp]),
        "rcg_episode_reset": (C.c_int, [vp]),
        "rcg_episode_stats": (C.c_int, [vp, i32, vp, C.POINTER(RcgSummary)]),
        "rcg_tick_count": (i64, [vp]),
        "rcg_set_tick_count": (C.c_int, [vp, i64]),
        "rcg_profile": (C.c_int, [vp, i32]),
        "rcg_profile_read": (C.c_int, [vp, i32, C.POINTER(C.c_double), C.POINTER(i64)]),
        "rcg_profile_samples": (C.c_int, [vp, i32, C.POINTER(C.c_double), i64, C.POINTER(i64)]),
        "rcg_last_launch": (C.c_int, [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "rcg_kernel_name": (C.c_char_p, [i32]),
        "rcg_wait_stream": (C.c_int, [vp, vp]),
        "rcg_release_stream": (C.c_int, [vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the .so does not export what rcg.h declares
        fn.restype = res
        fn.argtypes = args
    if L.rcg_version() != RCG_VERSION:
        raise ImportError(f"librcg.so version {L.rcg_version()} != binding version {RCG_VERSION}; rebuild")
    _lib = L
    return L


def last_error(handle=None) -> str:
    return lib().rcg_last_error(handle).decode("utf-8", "replace")


def check(rc, handle=None, allow=()):
    if rc != OK and rc not in allow:
        raise NativeError(rc, last_error(handle))
    return rc
