"""Engine: one librcg handle = a batch of B closed loops (System + Simulator + CtrlOptPred state)
resident in HBM on one MI355X.

Host-side arrays use the reference's natural shapes with a leading batch axis (``state [B, ds]``,
``action_sqn [B, K, N, du]``); the device layout is struct-of-arrays with the env index innermost
(include/rcg.h).  Inputs may be numpy arrays (uploaded), :class:`DeviceArray` or torch CUDA tensors
(zero-copy, must already be in the device layout and dtype).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _native as N


@dataclass
class EngineConfig:
    """Python mirror of ``rcg_cfg``.  Names follow the reference's constructor arguments
    (rcognita/systems.py:69-79, rcognita/controllers.py:811-837)."""

    sys_id: int
    batch: int = 1
    dtype: str = "f32"  # "f32" | "f64"
    device: int = 0
    Nactor: int = 5
    mode: str = "MPC"
    stage_obj_struct: str = "quadratic"
    critic_struct: str = "quad-nomix"
    Ncritic: int = 4
    buffer_size: int = 0  # 0: no critic buffers (MPC)
    substeps_per_tick: int = 1
    critic_every_ticks: int = 1  # critic_period / sampling_time (controllers.py:1466)
    dt_sim: float = 0.01
    sampling_time: float = 0.01
    pred_step_size: float = 0.02
    gamma: float = 1.0
    pars: Sequence[float] = ()
    ctrl_bnds: Optional[np.ndarray] = None  # [du, 2]
    R1: Optional[np.ndarray] = None  # [n, n] or diagonal [n]
    R2: Optional[np.ndarray] = None
    observation_target: Optional[Sequence[float]] = None  # None <=> the reference's []
    action_init: Optional[Sequence[float]] = None  # None <=> action_min / 10 (controllers.py:973-975)
    per_env_pars: bool = False
    ref_lag: bool = False
    accum_every_substep: bool = False
    # disturbance model, System(is_disturb=1, pars_disturb=[sigma, mu, tau]) (systems.py:303-306, 337)
    is_disturb: bool = False
    pars_disturb: Optional[Sequence] = None  # [sigma, mu, tau], each of length dim_disturb
    disturb_init: Optional[Sequence[float]] = None
    seed: int = 0            # key of the counter-based noise generator
    env_id_base: int = 0     # global id of this handle's env 0 (shards of one job share `seed`)

    def to_native(self) -> N.RcgCfg:
        ds, du, npar = N.SYS_DIMS[self.sys_id]
        n = ds + du
        c = N.RcgCfg()
        c.struct_size = C.sizeof(N.RcgCfg)
        c.sys_id, c.batch, c.device = int(self.sys_id), int(self.batch), int(self.device)
        c.dtype = {"f32": N.F32, "f64": N.F64}[self.dtype]
        c.n_actor = int(self.Nactor)
        c.mode = N.MODE_IDS[self.mode]
        c.stage_obj_struct = N.STAGE_IDS[self.stage_obj_struct]
        c.critic_struct = N.CRITIC_IDS[self.critic_struct]
        c.n_critic, c.buffer_size = int(self.Ncritic), int(self.buffer_size)
        c.substeps_per_tick = int(self.substeps_per_tick)
        c.critic_every_ticks = int(self.critic_every_ticks)
        c.dt_sim, c.sampling_time = float(self.dt_sim), float(self.sampling_time)
        c.pred_step_size, c.gamma = float(self.pred_step_size), float(self.gamma)
        flags = 0
        pars = np.asarray(self.pars, dtype=np.float64).reshape(-1)
        if len(pars) < npar:
            raise ValueError(f"system needs {npar} parameters, got {len(pars)}")
        for i in range(npar):
            c.pars[i] = pars[i]
        bnds = np.zeros((du, 2)) if self.ctrl_bnds is None else np.asarray(self.ctrl_bnds, dtype=np.float64).reshape(du, 2)
        for i in range(du):
            c.ctrl_bnds[2 * i], c.ctrl_bnds[2 * i + 1] = bnds[i, 0], bnds[i, 1]
        for name, M in (("R1", self.R1), ("R2", self.R2)):
            if M is None:
                M = np.eye(n) if name == "R1" else np.zeros((n, n))
            M = np.asarray(M, dtype=np.float64)
            if M.ndim == 1:
                M = np.diag(M)
            if M.shape != (n, n):
                raise ValueError(f"{name} must be [{n},{n}] (or its diagonal), got {M.shape}")
            arr = getattr(c, name)
            for i in range(n):
                for j in range(n):
                    arr[i * n + j] = M[i, j]
        if self.observation_target is not None and len(self.observation_target) > 0:
            t = np.asarray(self.observation_target, dtype=np.float64).reshape(ds)
            for i in range(ds):
                c.target[i] = t[i]
            flags |= N.FLAG_HAS_TARGET
        a0 = bnds[:, 0] / 10.0 if self.action_init is None or len(self.action_init) == 0 else np.asarray(
            self.action_init, dtype=np.float64).reshape(du)
        for i in range(du):
            c.action_init[i] = a0[i]
        dc = dim_critic(self.critic_struct, ds, du)
        lo, hi = critic_bounds(self.critic_struct)
        for i in range(dc):
            c.w_init[i], c.w_min[i], c.w_max[i] = 1.0, lo, hi  # controllers.py:1026-1042
        if self.per_env_pars:
            flags |= N.FLAG_PER_ENV_PARS
        if self.ref_lag:
            flags |= N.FLAG_REF_LAG
        if self.accum_every_substep:
            flags |= N.FLAG_ACCUM_EVERY_SUBSTEP
        if self.is_disturb:
            dd = N.DIM_DISTURB[int(self.sys_id)]
            flags |= N.FLAG_DISTURB
            if self.pars_disturb is None or len(self.pars_disturb) != 3:
                raise ValueError("is_disturb needs pars_disturb = [sigma, mu, tau]")
            for row, v in enumerate(self.pars_disturb):
                v = np.asarray(v, dtype=np.float64).reshape(-1)
                if len(v) < dd:
                    raise ValueError(f"pars_disturb[{row}] needs {dd} entries, got {len(v)}")
                for k in range(dd):
                    c.pars_disturb[2 * row + k] = v[k]
            if self.disturb_init is not None and len(self.disturb_init):
                q0 = np.asarray(self.disturb_init, dtype=np.float64).reshape(-1)
                for k in range(dd):
                    c.disturb_init[k] = q0[k]
        # the counter-based generators (disturbance noise, candidate search) are keyed by (seed, global env id)
        c.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF
        c.env_id_base = int(self.env_id_base)
        c.flags = flags
        return c


def dim_critic(critic_struct: str, dy: int, du: int) -> int:
    """rcognita/controllers.py:1024-1039."""
    n = dy + du
    return {"quad-lin": n * (n + 1) // 2 + n, "quadratic": n * (n + 1) // 2, "quad-nomix": n,
            "quad-mix": dy + dy * du + du}[critic_struct]


def critic_bounds(critic_struct: str):
    """(Wmin, Wmax) scalars, rcognita/controllers.py:1026-1039."""
    return (-1e3, 1e3) if critic_struct in ("quad-lin", "quad-mix") else (0.0, 1e3)


class DeviceArray:
    """A raw HBM allocation made through the C ABI (no torch needed).

    ``scratch=True`` marks the engine's own temporaries (uploaded inputs and output buffers of one call): small ones are
    returned to a per-engine free list instead of hipFree'd - at B = 1 the drop-in classes made 12 allocations per
    simulation step, a fifth of the step's wall time.  Safe because a scratch array is only ever touched on its engine's
    stream (rcg_memcpy_* are stream-ordered), so a reused buffer is written after the kernel that last read it."""

    POOL_MAX_ARRAY = 1 << 20   # larger temporaries go back to the driver at once
    POOL_MAX_TOTAL = 16 << 20  # per engine

    def __init__(self, engine: "Engine", shape, dtype, scratch=False):
        self.engine = engine
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        n = 1
        for v in self.shape:
            n *= v
        self.nbytes = n * self.dtype.itemsize
        self.scratch = bool(scratch)
        self._cap = 0
        if self.scratch and self.nbytes <= self.POOL_MAX_ARRAY:
            cap = 256
            while cap < self.nbytes:
                cap <<= 1
            self._cap = cap
            free = engine._pool.get(cap)
            if free:
                self.ptr = free.pop()
                engine._pool_bytes -= cap
                return
        p = C.c_void_p()
        N.check(N.lib().rcg_dev_alloc(engine._h, max(self._cap, self.nbytes, 16), C.byref(p)), engine._h)
        self.ptr = p.value

    def upload(self, arr):
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        if self.nbytes:
            N.check(N.lib().rcg_memcpy_h2d(self.engine._h, self.ptr, a.ctypes.data, self.nbytes), self.engine._h)
        return self

    def to_host(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            N.check(N.lib().rcg_memcpy_d2h(self.engine._h, out.ctypes.data, self.ptr, self.nbytes), self.engine._h)
        return out

    def rows(self, a: int, b: int) -> "DeviceArray":
        """Rows ``[a, b)`` along the first axis as a view (pointer offset; the view owns nothing and must not outlive
        this array)."""
        a, b = int(a), int(b)
        if not (0 <= a <= b <= self.shape[0]):
            raise IndexError(f"rows [{a}, {b}) of an array with {self.shape[0]}")
        v = DeviceArray.__new__(DeviceArray)
        v.engine, v.dtype, v.scratch, v._cap = self.engine, self.dtype, False, 0
        v.shape = (b - a,) + self.shape[1:]
        row_bytes = self.nbytes // self.shape[0] if self.shape[0] else 0
        v.nbytes = (b - a) * row_bytes
        v.ptr = (self.ptr + a * row_bytes) if self.ptr else None
        v._view_of = self
        return v

    def free(self):
        if getattr(self, "_view_of", None) is not None:  # a view: nothing to release
            self.ptr = None
            return
        e = self.engine
        if self.ptr and e._h:
            if self._cap and e._pool_bytes + self._cap <= self.POOL_MAX_TOTAL:
                e._pool.setdefault(self._cap, []).append(self.ptr)
                e._pool_bytes += self._cap
            else:
                N.lib().rcg_dev_free(e._h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


class Engine:
    """Owner of one ``rcg_handle``."""

    def __init__(self, cfg: EngineConfig):
        self.cfg = cfg
        self._h = None
        self._pool, self._pool_bytes = {}, 0  # free list of small scratch allocations, by capacity (DeviceArray)
        L = N.lib()
        self.ds, self.du, self.npar = N.SYS_DIMS[cfg.sys_id]
        self.dd = N.DIM_DISTURB[int(cfg.sys_id)]
        self.dy = self.ds
        self.B = int(cfg.batch)
        self.N = int(cfg.Nactor)
        self.real = np.float32 if cfg.dtype == "f32" else np.float64
        self.dc = dim_critic(cfg.critic_struct, self.dy, self.du)
        self.buffer_size = int(cfg.buffer_size)
        native = cfg.to_native()
        h = C.c_void_p()
        rc = L.rcg_create(C.byref(native), C.byref(h))
        if rc != N.OK:
            raise N.NativeError(rc, N.last_error(None))
        self._h = h
        self._loop_fn = L.rcg_loop_step
        self._loop_begin_fn, self._loop_end_fn = L.rcg_loop_step_begin, L.rcg_loop_step_end

    # ------------------------------------------------------------------ life cycle
    def close(self):
        if self._h:
            for ptrs in self._pool.values():
                for p in ptrs:
                    N.lib().rcg_dev_free(self._h, p)
            self._pool, self._pool_bytes = {}, 0
            N.lib().rcg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: Optional[int]):
        N.check(N.lib().rcg_set_stream(self._h, C.c_void_p(stream_ptr or 0)), self._h)

    def use_own_stream(self):
        """A non-blocking HIP stream owned by the handle (rcg_use_own_stream).  Nothing orders that stream against the
        stream a device-resident input (torch tensor, DeviceArray of another engine) was WRITTEN on: call
        :meth:`wait_stream` with the producer's stream before the first call that reads such an input."""
        N.check(N.lib().rcg_use_own_stream(self._h), self._h)

    def synchronize(self):
        N.check(N.lib().rcg_synchronize(self._h), self._h)

    def wait_stream(self, producer_stream_ptr: Optional[int]):
        """Order this handle's next launches after everything queued so far on another HIP stream (rcg_wait_stream):
        needed when a device-resident input was written on a stream other than the handle's - e.g. torch's current
        stream while the handle runs on a stream of its own.  No host synchronisation."""
        N.check(N.lib().rcg_wait_stream(self._h, C.c_void_p(producer_stream_ptr or 0)), self._h)

    def release_stream(self, consumer_stream_ptr: Optional[int]):
        """The reverse edge of :meth:`wait_stream` (rcg_release_stream): work queued on ``consumer_stream`` from now on
        waits for everything this handle has launched so far - before another stream overwrites or frees a device-resident
        input this handle's kernels may still be reading.  No host synchronisation."""
        N.check(N.lib().rcg_release_stream(self._h, C.c_void_p(consumer_stream_ptr or 0)), self._h)

    # ------------------------------------------------------------------ device memory
    def empty(self, shape, dtype=None) -> DeviceArray:
        return DeviceArray(self, shape, self.real if dtype is None else dtype)

    def _tmp(self, shape, dtype=None) -> DeviceArray:
        """Output buffer of ONE call of this engine (pooled, see DeviceArray)."""
        return DeviceArray(self, shape, self.real if dtype is None else dtype, scratch=True)

    def to_device(self, arr, dtype=None) -> DeviceArray:
        a = np.ascontiguousarray(arr, dtype=self.real if dtype is None else dtype)
        return DeviceArray(self, a.shape, a.dtype).upload(a)

    def _check_device_input(self, x, shape, dtype, what):
        """A device-resident input goes to the kernels as a bare pointer: refuse anything whose element type, shape,
        size or device differs from what the kernel will read (a float64 tensor on an f32 handle computes garbage, a
        short one makes the kernel read past the allocation and faults the GPU)."""
        dtype = np.dtype(dtype)
        shape = tuple(int(v) for v in shape)
        need = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if isinstance(x, DeviceArray):
            if x.engine.cfg.device != self.cfg.device:
                raise ValueError(f"{what}: DeviceArray lives on device {x.engine.cfg.device}, the handle on {self.cfg.device}")
            if x.dtype != dtype:
                raise ValueError(f"{what}: dtype {x.dtype} != {dtype} of this handle")
            if x.ptr is None:
                raise ValueError(f"{what}: DeviceArray was freed")
            have, xshape = x.nbytes, x.shape
        else:
            if not x.is_cuda or not x.is_contiguous():
                raise ValueError(f"{what}: torch inputs must be contiguous CUDA tensors in the device layout")
            if x.device.index != self.cfg.device:
                raise ValueError(f"{what}: tensor is on cuda:{x.device.index}, the handle on device {self.cfg.device}")
            tname = str(x.dtype).replace("torch.", "")
            if tname != dtype.name:
                raise ValueError(f"{what}: dtype {tname} != {dtype.name} of this handle")
            have, xshape = x.numel() * x.element_size(), tuple(x.shape)
        if tuple(xshape) != shape:
            raise ValueError(f"{what}: shape {tuple(xshape)} != expected device layout {shape}")
        if have < need:
            raise ValueError(f"{what}: {have} bytes, the kernel reads {need}")
        return C.c_void_p(x.ptr if isinstance(x, DeviceArray) else x.data_ptr())

    def _in(self, x, keep, soa_from=None, dev_shape=None, what="input", dtype=None):
        """Device pointer of an input.  numpy -> (optionally AoS->SoA transposed) upload; device-resident inputs
        (DeviceArray, torch) must already have the device layout ``dev_shape`` and the handle's element type."""
        if x is None:
            return None
        if isinstance(x, DeviceArray) or _is_torch(x):
            if dev_shape is None:
                raise ValueError(f"{what}: device-resident inputs are not accepted here")
            return self._check_device_input(x, dev_shape, self.real if dtype is None else dtype, what)
        a = np.asarray(x, dtype=self.real)
        if soa_from is not None:
            a = soa_from(a)
        a = np.ascontiguousarray(a, dtype=self.real)
        d = DeviceArray(self, a.shape, a.dtype, scratch=True).upload(a)
        keep.append(d)
        return C.c_void_p(d.ptr)

    def _soa_item(self, x, d, what):
        """:meth:`_in_many` item for a per-env input ``[B, d]`` on the host / ``[d, B]`` on the device."""
        return (x, lambda a: a.reshape(self.B, d).T, (d, self.B), what)

    def _in_soa(self, x, keep, d, what):
        """Per-env input ``[B, d]`` on the host, or ``[d, B]`` already on the device."""
        return self._in(x, keep, lambda a: a.reshape(self.B, d).T, dev_shape=(d, self.B), what=what)

    # One host <-> device copy per direction and call: at B = 1 a synchronous copy costs 15-30 us whatever its size, and
    # the reference's loop makes a dozen of them per simulation step through these methods.
    def _in_many(self, items, keep):
        """``items``: list of ``(x, soa_from, dev_shape, what)`` as for :meth:`_in`; host arrays travel in ONE upload."""
        ptrs, host, total = [None] * len(items), [], 0
        for i, (x, soa_from, dev_shape, what) in enumerate(items):
            if x is None:
                continue
            if isinstance(x, DeviceArray) or _is_torch(x):
                ptrs[i] = self._in(x, keep, soa_from, dev_shape, what)
                continue
            a = np.asarray(x, dtype=self.real)
            if soa_from is not None:
                a = soa_from(a)
            a = np.ascontiguousarray(a)
            host.append((i, total, a))
            total += (a.nbytes + 255) & ~255
        if host:
            if len(host) == 1:
                i, _, a = host[0]
                d = DeviceArray(self, a.shape, a.dtype, scratch=True).upload(a)
                ptrs[i] = C.c_void_p(d.ptr)
            else:
                blob = np.zeros(total, dtype=np.uint8)
                for i, o, a in host:
                    blob[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
                d = DeviceArray(self, (total,), np.uint8, scratch=True).upload(blob)
                for i, o, a in host:
                    ptrs[i] = C.c_void_p(d.ptr + o)
            keep.append(d)
        return ptrs

    def _out_many(self, specs):
        """``specs``: list of ``(shape, dtype)`` (None entries: output not wanted) -> (pointers, fetch): device buffers
        carved from ONE scratch allocation; ``fetch()`` brings them back in ONE copy as host arrays."""
        offs, total = [], 0
        for sp in specs:
            if sp is None:
                offs.append(None)
                continue
            shape, dt = sp
            n = int(np.prod(shape, dtype=np.int64)) * np.dtype(self.real if dt is None else dt).itemsize
            offs.append((total, n))
            total += (n + 255) & ~255
        blob = DeviceArray(self, (max(total, 16),), np.uint8, scratch=True)
        ptrs = [None if o is None else C.c_void_p(blob.ptr + o[0]) for o in offs]

        def fetch():
            h = blob.to_host()
            out = []
            for sp, o in zip(specs, offs):
                if sp is None:
                    out.append(None)
                    continue
                shape, dt = sp
                out.append(h[o[0]:o[0] + o[1]].view(self.real if dt is None else dt).reshape(shape).copy())
            return out

        return ptrs, fetch

    # ------------------------------------------------------------------ per-env tensors
    _FIELD_DIMS = {
        N.FIELD_STATE: "ds", N.FIELD_ACTION: "du", N.FIELD_ACCUM: None, N.FIELD_STEP_IDX: None,
        N.FIELD_EPISODE_IDX: None, N.FIELD_STATUS: None, N.FIELD_PARS: "npar", N.FIELD_STATE_INIT: "ds",
        N.FIELD_STATE_PREV: "ds", N.FIELD_BEST_J: None, N.FIELD_BEST_IDX: None, N.FIELD_W_CRITIC: "dc",
        N.FIELD_W_PREV: "dc", N.FIELD_RETURNS: None, N.FIELD_DISTURB: "dd", N.FIELD_SUBSTEP_IDX: None,
    }
    _FIELD_INT = {N.FIELD_STEP_IDX: np.int32, N.FIELD_EPISODE_IDX: np.int32, N.FIELD_BEST_IDX: np.int32,
                  N.FIELD_STATUS: np.uint32, N.FIELD_SUBSTEP_IDX: np.int32}

    def _field_meta(self, f):
        """(device shape, dtype, host->device transform, device->host transform)."""
        dt = self._FIELD_INT.get(f, self.real)
        B = self.B
        if f == N.FIELD_ACTION_SQN:  # one row per env, the reference's flat step-major layout
            return (B, self.N, self.du), dt, (lambda a: a.reshape(B, self.N, self.du)), (lambda a: a)
        if f in (N.FIELD_OBS_BUF, N.FIELD_ACT_BUF):
            d = self.dy if f == N.FIELD_OBS_BUF else self.du
            return (self.buffer_size, d, B), dt, (lambda a: a.reshape(B, self.buffer_size, d).transpose(1, 2, 0)), (
                lambda a: a.transpose(2, 0, 1))
        dim = self._FIELD_DIMS[f]
        if dim is None:
            return (B,), dt, (lambda a: a.reshape(B)), (lambda a: a)
        d = getattr(self, dim)
        return (d, B), dt, (lambda a: np.broadcast_to(a, (B, d)).T), (lambda a: a.T)

    def set_field(self, f, value):
        """Host array in the reference's shape (``[B, d]``, ``[B]``, ``[B, buffer_size, d]``)."""
        shape, dt, h2d, _ = self._field_meta(f)
        if isinstance(value, DeviceArray) or _is_torch(value):
            ptr = self._check_device_input(value, shape, dt, f"set_field({f})")  # rcg_set_field copies the whole field
            N.check(N.lib().rcg_set_field(self._h, f, ptr, N.DEVICE), self._h)
            return
        a = np.ascontiguousarray(h2d(np.asarray(value, dtype=dt)), dtype=dt)
        assert a.shape == shape, (a.shape, shape)
        N.check(N.lib().rcg_set_field(self._h, f, a.ctypes.data, N.HOST), self._h)

    def get_field(self, f) -> np.ndarray:
        shape, dt, _, d2h = self._field_meta(f)
        out = np.empty(shape, dtype=dt)
        N.check(N.lib().rcg_get_field(self._h, f, out.ctypes.data, N.HOST), self._h)
        return np.ascontiguousarray(d2h(out))

    def field_ptr(self, f) -> int:
        p = C.c_void_p()
        N.check(N.lib().rcg_field_ptr(self._h, f, C.byref(p)), self._h)
        return p.value

    def set_state(self, state, also_init=True):
        self.set_field(N.FIELD_STATE, state)
        if also_init:
            self.set_field(N.FIELD_STATE_INIT, state)

    def get_state(self):
        return self.get_field(N.FIELD_STATE)

    # ------------------------------------------------------------------ stateless operators
    def rhs(self, state, action, clip=False):
        """``_state_dyn`` (clip=False) / ``closed_loop_rhs`` (clip=True) on ``n`` points.
        Returns ``(dstate [n, ds], clipped_action [n, du])``."""
        state = np.asarray(state, dtype=self.real).reshape(-1, self.ds)
        action = np.asarray(action, dtype=self.real).reshape(-1, self.du)
        n = state.shape[0]
        keep = []
        ps = self._in(state, keep, lambda a: a.T)
        pa = self._in(action, keep, lambda a: a.T)
        d, ca = self._tmp((self.ds, n)), self._tmp((self.du, n))
        N.check(N.lib().rcg_rhs(self._h, ps, pa, C.c_void_p(d.ptr), C.c_void_p(ca.ptr), n, 1 if clip else 0), self._h)
        return d.to_host().T.copy(), ca.to_host().T.copy()

    def rhs_full(self, state, disturb, action, xi, clip=True):
        """``closed_loop_rhs`` on the full state of a system with ``is_disturb=1`` for ``n`` points, the noise value
        ``xi [n, dd]`` given (rcg_rhs_full).  Returns ``(dstate [n, ds], ddisturb [n, dd], clipped_action [n, du])``."""
        state = np.asarray(state, dtype=self.real).reshape(-1, self.ds)
        disturb = np.asarray(disturb, dtype=self.real).reshape(-1, self.dd)
        action = np.asarray(action, dtype=self.real).reshape(-1, self.du)
        xi = np.asarray(xi, dtype=self.real).reshape(-1, self.dd)
        n = state.shape[0]
        keep = []
        ptrs = [self._in(a, keep, lambda v: v.T) for a in (state, disturb, action, xi)]
        d, dq, ca = self._tmp((self.ds, n)), self._tmp((self.dd, n)), self._tmp((self.du, n))
        N.check(N.lib().rcg_rhs_full(self._h, *ptrs, C.c_void_p(d.ptr), C.c_void_p(dq.ptr), C.c_void_p(ca.ptr), n,
                                     1 if clip else 0), self._h)
        return d.to_host().T.copy(), dq.to_host().T.copy(), ca.to_host().T.copy()

    def disturb_noise(self):
        """What the next ``sim_step`` substep will draw: ``(bits [B, 4] uint32, xi [B, 2])`` (rcg_disturb_noise)."""
        bits, xi = self._tmp((4, self.B), np.uint32), self._tmp((2, self.B))
        N.check(N.lib().rcg_disturb_noise(self._h, C.c_void_p(bits.ptr), C.c_void_p(xi.ptr)), self._h)
        return bits.to_host().T.copy(), xi.to_host().T.copy()

    def stage_obj(self, obs, act):
        obs = np.asarray(obs, dtype=self.real).reshape(-1, self.dy)
        act = np.asarray(act, dtype=self.real).reshape(-1, self.du)
        n = obs.shape[0]
        keep = []
        po, pa = self._in_many([(obs, lambda a: a.T, None, "obs"), (act, lambda a: a.T, None, "act")], keep)
        (pout,), fetch = self._out_many([((n,), None)])
        N.check(N.lib().rcg_stage_obj(self._h, po, pa, pout, n), self._h)
        return fetch()[0]

    def critic(self, obs, act, w):
        obs = np.asarray(obs, dtype=self.real).reshape(-1, self.dy)
        act = np.asarray(act, dtype=self.real).reshape(-1, self.du)
        w = np.asarray(w, dtype=self.real).reshape(-1, self.dc)
        n = obs.shape[0]
        keep = []
        po, pa, pw = self._in_many([(obs, lambda a: a.T, None, "obs"), (act, lambda a: a.T, None, "act"),
                                    (w, lambda a: a.T, None, "w")], keep)
        (pout,), fetch = self._out_many([((n,), None)])
        N.check(N.lib().rcg_critic(self._h, po, pa, pw, pout, n), self._h)
        return fetch()[0]

    def _cand(self, cand, keep, K=None):
        """Candidates ``[B, K, N, du]`` (numpy / device).  Returns (pointer, K).  An explicit ``K`` must agree with the
        tensor: the kernels index ``B*K*N*du`` elements behind the pointer."""
        if cand is None:
            return None, (None if K is None else int(K))
        if isinstance(cand, DeviceArray) or _is_torch(cand):
            shape = tuple(int(v) for v in cand.shape)
            if len(shape) != 4:
                raise ValueError(f"candidates on the device must be [B, K, N, du], got shape {shape}")
            Kc = shape[1]
            ptr = self._check_device_input(cand, (self.B, Kc, self.N, self.du), self.real, "candidates")
        else:
            a = np.asarray(cand, dtype=self.real)
            if a.ndim == 3:  # [K, N, du] shared by all envs
                a = np.broadcast_to(a[None], (self.B,) + a.shape)
            a = a.reshape(self.B, -1, self.N, self.du)
            ptr, Kc = self._in(a, keep), a.shape[1]
        if K is not None and int(K) != Kc:
            raise ValueError(f"K = {K} given, but the candidate tensor holds {Kc} sequences per env")
        return ptr, Kc

    def actor_cost(self, cand, obs=None, state_sys=None, w=None):
        """``_actor_cost`` of every candidate: ``cand [B, K, N, du]`` -> ``J [B, K]``."""
        keep = []
        pc, K = self._cand(cand, keep)
        J = self._tmp((self.B, K))
        N.check(N.lib().rcg_actor_cost(self._h, pc, K, self._in_soa(obs, keep, self.dy, "obs"),
                                       self._in_soa(state_sys, keep, self.ds, "state_sys"),
                                       self._in_soa(w, keep, self.dc, "w"), C.c_void_p(J.ptr)), self._h)
        return J.to_host()

    def critic_cost(self, w=None):
        keep = []
        Jc = self._tmp((self.B,))
        N.check(N.lib().rcg_critic_cost(self._h, self._in_soa(w, keep, self.dc, "w"), C.c_void_p(Jc.ptr)), self._h)
        return Jc.to_host()

    # ------------------------------------------------------------------ stateful steps
    def set_tick_parts(self, parts):
        """0 = automatic (an eligible RQL / SQL tick is split into two halves on two internal streams from 65 536 envs),
        1 = never, 2 = whenever eligible (rcg_set_tick_parts)."""
        N.check(N.lib().rcg_set_tick_parts(self._h, int(parts)), self._h)

    def join(self):
        """Order the handle's stream behind the halves of a split tick, without a host wait (rcg_join)."""
        N.check(N.lib().rcg_join(self._h), self._h)

    def sim_step(self, n_substeps=1, step=None):
        """``n_substeps`` RK4 substeps of the handle's ``dt_sim``; with ``step``: ONE step of that length, cut into
        ``n_substeps`` substeps (rcg_sim_step_h)."""
        if step is None:
            N.check(N.lib().rcg_sim_step(self._h, int(n_substeps)), self._h)
        else:
            N.check(N.lib().rcg_sim_step_h(self._h, int(n_substeps), float(step)), self._h)

    def loop_step(self, action, step, n_substeps=1, decide=False, push=False, fit=False, iters=10):
        """One iteration of the reference's loop body in one native call and one host wait (rcg_loop_step): hold ``action
        [B, du]`` (None: the handle's own) over one simulation step of length ``step``, then - as asked - push the critic
        buffers, refit the critic, decide with the on-device optimiser (rollout from the state before the step), and
        evaluate the stage cost of (new state, action).  Returns ``(state [B, ds], action [B, du], stage_obj [B], best_J [B]
        (NaN without ``decide``), w_critic [B, dc] or None)`` as float64."""
        ds, du = self.ds, self.du
        row = ds + du + 2 + (self.dc if self.cfg.mode != "MPC" else 0)
        out = np.empty((self.B, row), dtype=np.float64)  # (fresh per call: the caller keeps views of it)
        pa = None
        if action is not None:
            a = np.asarray(action, dtype=np.float64)
            if a.shape != (self.B, du) or not a.flags.c_contiguous:
                a = np.ascontiguousarray(np.broadcast_to(a, (self.B, du)))
            pa = a.ctypes.data
        flags = (N.LOOP_DECIDE if decide else 0) | (N.LOOP_PUSH if push else 0) | (N.LOOP_FIT if fit else 0)
        rc = self._loop_fn(self._h, pa, step, n_substeps, flags, iters, out.ctypes.data)
        if rc:
            N.check(rc, self._h)
        return (out[:, :ds], out[:, ds:ds + du], out[:, ds + du], out[:, ds + du + 1],
                out[:, ds + du + 2:] if row > ds + du + 2 else None)

    def loop_step_begin(self, action, step, n_substeps=1, decide=False, push=False, fit=False, iters=10):
        """Enqueue one loop iteration and return (rcg_loop_step_begin): the device runs it while the host goes on;
        ``loop_step_end`` collects it.  One iteration may be pending."""
        pa = None
        if action is not None:
            a = np.asarray(action, dtype=np.float64)
            if a.shape != (self.B, self.du) or not a.flags.c_contiguous:
                a = np.ascontiguousarray(np.broadcast_to(a, (self.B, self.du)))
            pa = a.ctypes.data  # (copied into the handle's pinned buffer before the call returns)
        flags = (N.LOOP_DECIDE if decide else 0) | (N.LOOP_PUSH if push else 0) | (N.LOOP_FIT if fit else 0)
        rc = self._loop_begin_fn(self._h, pa, step, n_substeps, flags, iters)
        if rc:
            N.check(rc, self._h)

    def loop_step_end(self, drop=False):
        """Wait for the pending iteration (rcg_loop_step_end) and return its rows as ``loop_step`` does; ``drop``: wait only."""
        if drop:
            rc = self._loop_end_fn(self._h, None)
            if rc:
                N.check(rc, self._h)
            return None
        ds, du = self.ds, self.du
        row = ds + du + 2 + (self.dc if self.cfg.mode != "MPC" else 0)
        out = np.empty((self.B, row), dtype=np.float64)  # (fresh per call: the caller keeps views of it)
        rc = self._loop_end_fn(self._h, out.ctypes.data)
        if rc:
            N.check(rc, self._h)
        return (out[:, :ds], out[:, ds:ds + du], out[:, ds + du], out[:, ds + du + 1],
                out[:, ds + du + 2:] if row > ds + du + 2 else None)

    def actor_argmin(self, cand=None, K=None, obs=None, state_sys=None):
        """Returns ``(action [B, du], best_J [B], best_idx [B] int32)``."""
        keep = []
        pc, K = self._cand(cand, keep, K)
        if K is None:
            raise ValueError("actor_argmin: K is required with generated candidates (cand=None)")
        pobs, pxs = self._in_many([self._soa_item(obs, self.dy, "obs"), self._soa_item(state_sys, self.ds, "state_sys")], keep)
        (pact, pbj, pbi), fetch = self._out_many([((self.du, self.B), None), ((self.B,), None), ((self.B,), np.int32)])
        N.check(N.lib().rcg_actor_argmin(self._h, pc, K, pobs, pxs, pact, pbj, pbi), self._h)
        act, bj, bi = fetch()
        return act.T.copy(), bj, bi

    def control_tick(self, cand=None, K=None, T=1):
        """One env.control-step for all envs (``T`` > 1: T of them with the same candidates in one native call,
        rcg_control_tick_n - at small batches the Python round trip is longer than the tick and lets the GPU idle and
        clock down).  ``cand`` on device for
        the timed path."""
        keep = []
        pc, K = self._cand(cand, keep, K)
        if K is None:
            raise ValueError("control_tick: K is required with generated candidates (cand=None)")
        if int(T) == 1:
            N.check(N.lib().rcg_control_tick(self._h, pc, K), self._h)
        else:
            N.check(N.lib().rcg_control_tick_n(self._h, pc, K, int(T)), self._h)
        if keep:  # temporaries were uploaded for this call: finish before they are freed
            self.synchronize()

    def control_ticks(self, T, K):
        """``T`` env.control-steps with the generated candidate grid in ONE launch (rcg_control_ticks): bit-identical
        to ``T`` calls of ``control_tick(None, K)``.  MPC with any stage cost, with or without the disturbance model; RQL /
        SQL whose tick fits k_ticks_mem (rcg.h); other handles are refused (RCG_ERR_UNSUPPORTED) and loop control_tick."""
        N.check(N.lib().rcg_control_ticks(self._h, int(T), int(K)), self._h)

    def actor_optimize(self, iters=10, obs=None, state_sys=None, u_init=None):
        """On-device actor optimiser (rcg_actor_optimize), every mode and cost structure: adjoint gradient, limited-memory
        quasi-Newton direction on the free coordinates, 16-way projected line search; RQL / SQL use the handle's
        ``W_CRITIC``.  ``u_init [B, N, du]`` (None: the reference's ``action_sqn_init``).  Returns
        ``(action [B, du], u_opt [B, N, du], best_J [B], n_iter [B] int32)``."""
        keep = []
        u0 = None if u_init is None else np.broadcast_to(
            np.asarray(u_init, dtype=self.real).reshape(-1, self.N, self.du), (self.B, self.N, self.du))
        pobs, pxs, pu = self._in_many([self._soa_item(obs, self.dy, "obs"), self._soa_item(state_sys, self.ds, "state_sys"),
                                       (u0, None, None, "u_init")], keep)
        (puo, pact, pbj, pni), fetch = self._out_many([((self.B, self.N, self.du), None), ((self.du, self.B), None),
                                                       ((self.B,), None), ((self.B,), np.int32)])
        N.check(N.lib().rcg_actor_optimize(self._h, int(iters), pobs, pxs, pu, puo, pact, pbj, pni), self._h)
        uo, act, bj, ni = fetch()
        return act.T.copy(), uo, bj, ni

    def control_tick_opt(self, iters=10, warm_start=False):
        """One env.control-step with the on-device optimiser as the decision (rcg_control_tick_opt); RQL / SQL: the
        buffer push and the critic fit come between the env step and the decision, as in ``control_tick``."""
        N.check(N.lib().rcg_control_tick_opt(self._h, int(iters), 1 if warm_start else 0), self._h)

    def actor_search(self, K=256, rounds=6, obs=None, state_sys=None, centre=None):
        """Device-side candidate search (rcg_actor_search): ``rounds`` rounds of ``K`` candidates generated, evaluated and
        refined in one launch; ``centre [B, N, du]`` (None: the reference's ``action_sqn_init``).  Returns
        ``(action [B, du], u_best [B, N, du], best_J [B], best_idx [B] int32)``."""
        keep = []
        c0 = None if centre is None else np.broadcast_to(
            np.asarray(centre, dtype=self.real).reshape(-1, self.N, self.du), (self.B, self.N, self.du))
        pobs, pxs, pc = self._in_many([self._soa_item(obs, self.dy, "obs"), self._soa_item(state_sys, self.ds, "state_sys"),
                                       (c0, None, None, "centre")], keep)
        (pub, pact, pbj, pbi), fetch = self._out_many([((self.B, self.N, self.du), None), ((self.du, self.B), None),
                                                       ((self.B,), None), ((self.B,), np.int32)])
        N.check(N.lib().rcg_actor_search(self._h, int(K), int(rounds), pobs, pxs, pc, pub, pact, pbj, pbi), self._h)
        ub, act, bj, bi = fetch()
        return act.T.copy(), ub, bj, bi

    def control_tick_search(self, K=256, rounds=6, warm_start=False):
        """One env.control-step with the device-side candidate search as the decision (rcg_control_tick_search)."""
        N.check(N.lib().rcg_control_tick_search(self._h, int(K), int(rounds), 1 if warm_start else 0), self._h)

    def candidates_sample(self, K, round=0, centre=None, out=None):
        """The candidate rows of one search round (rcg_candidates_sample) -> host ``[B, K, N, du]``, or, with ``out`` (a
        DeviceArray / torch tensor of that shape), written there and left on the device."""
        keep = []
        c0 = None if centre is None else np.broadcast_to(
            np.asarray(centre, dtype=self.real).reshape(-1, self.N, self.du), (self.B, self.N, self.du))
        (pc,) = self._in_many([(c0, None, (self.B, self.N, self.du), "centre")], keep)
        shape = (self.B, int(K), self.N, self.du)
        if out is not None:
            po = self._check_device_input(out, shape, self.real, "out")
            N.check(N.lib().rcg_candidates_sample(self._h, po, int(K), int(round), pc), self._h)
            if keep:
                self.synchronize()
            return out
        d = self._tmp(shape)
        N.check(N.lib().rcg_candidates_sample(self._h, C.c_void_p(d.ptr), int(K), int(round), pc), self._h)
        return d.to_host()

    def set_optimizer(self, memory=-1, ftol=None):
        """Curvature pairs the optimiser keeps per env (rcg_set_optimizer): 0 = projected steepest descent .. 8; -1 = the
        default (4 for RQL / SQL and non-diagonal stage costs, 0 for MPC with a diagonal R1).  ``ftol`` (rcg_set_optimizer_tol):
        an env is done after an accepted step that lowered J by no more than that; 0 = no such test (a handle's default)."""
        N.check(N.lib().rcg_set_optimizer(self._h, int(memory)), self._h)
        if ftol is not None:
            N.check(N.lib().rcg_set_optimizer_tol(self._h, float(ftol)), self._h)

    @staticmethod
    def _ctrl_pars(ctrl_pars):
        if ctrl_pars is None:
            return None
        m, I = (float(v) for v in ctrl_pars)
        return (C.c_double * 2)(m, I)

    def nominal_action(self, obs, ctrl_gain, ctrl_pars=None, clip=False, want_lyap=False):
        """Nominal controller of the handle's system on ``obs [n, dy]`` (rcg_nominal_action): returns ``action [n, du]``
        (and the Lyapunov value ``[n]`` with ``want_lyap``).  ``clip`` = the clip of ``compute_action``."""
        obs = np.asarray(obs, dtype=self.real).reshape(-1, self.dy)
        n = obs.shape[0]
        keep = []
        (pact, plyap), fetch = self._out_many([((self.du, n), None), ((n,), None) if want_lyap else None])
        N.check(N.lib().rcg_nominal_action(self._h, self._in(obs, keep, lambda a: a.T), pact, plyap, n, float(ctrl_gain),
                                           self._ctrl_pars(ctrl_pars), 1 if clip else 0), self._h)
        act, lyap = fetch()
        a = act.T.copy()
        return (a, lyap) if want_lyap else a

    def nominal_theta(self, obs):
        """theta* of CtrlNominal3WRobot's search for ``obs [n, dy]`` (rcg_nominal_theta) -> ``[n]``."""
        obs = np.asarray(obs, dtype=self.real).reshape(-1, self.dy)
        n = obs.shape[0]
        keep = []
        (pth,), fetch = self._out_many([((n,), None)])
        N.check(N.lib().rcg_nominal_theta(self._h, self._in(obs, keep, lambda a: a.T), pth, n), self._h)
        return fetch()[0]

    def control_tick_nominal(self, ctrl_gain, ctrl_pars=None):
        """One env.control-step with the nominal controller as the decision (rcg_control_tick_nominal)."""
        N.check(N.lib().rcg_control_tick_nominal(self._h, float(ctrl_gain), self._ctrl_pars(ctrl_pars)), self._h)

    def critic_update(self, do_fit=True):
        N.check(N.lib().rcg_critic_update(self._h, 1 if do_fit else 0), self._h)

    def episode_reset(self):
        N.check(N.lib().rcg_episode_reset(self._h), self._h)

    def episode_stats(self, from_accum=False, want_returns=False):
        """Returns ``(summary dict, returns [B] or None)``; ``n_failed > 0`` does not raise here."""
        s = N.RcgSummary()
        out = np.empty((self.B,), dtype=self.real) if want_returns else None
        N.check(N.lib().rcg_episode_stats(self._h, 1 if from_accum else 0, out.ctypes.data if want_returns else None,
                                          C.byref(s)), self._h, allow=(N.ERR_NONFINITE,))
        return {k: getattr(s, k) for k in ("count", "sum", "sumsq", "min", "max", "n_failed")}, out

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self):
        """Everything the handle carries between ticks: every allocated per-env tensor (host arrays in the reference's
        shapes) plus the episode's tick counter (rcg_tick_count)."""
        out = {"tick_count": np.int64(N.lib().rcg_tick_count(self._h)), "batch": np.int64(self.B),
               "sys_id": np.int64(self.cfg.sys_id), "dtype": np.str_(self.cfg.dtype)}
        for f in range(N.FIELD_COUNT):
            if N.lib().rcg_field_bytes(self._h, f) > 0:
                out[f"field_{f}"] = self.get_field(f)
        return out

    def load_state_dict(self, sd):
        if int(sd["batch"]) != self.B or int(sd["sys_id"]) != int(self.cfg.sys_id) or str(sd["dtype"]) != self.cfg.dtype:
            raise ValueError("checkpoint was written by a handle with another batch / system / dtype")
        for f in range(N.FIELD_COUNT):
            have = N.lib().rcg_field_bytes(self._h, f) > 0
            if have != (f"field_{f}" in sd):
                raise ValueError(f"checkpoint and handle disagree on field {f} (different mode / flags)")
        # STATE first: rcg_set_field(STATE) also resets STATE_PREV, which is then restored from its own entry
        order = [N.FIELD_STATE] + [f for f in range(N.FIELD_COUNT) if f != N.FIELD_STATE]
        for f in order:
            if f"field_{f}" in sd:
                self.set_field(f, sd[f"field_{f}"])
        N.check(N.lib().rcg_set_tick_count(self._h, int(sd["tick_count"])), self._h)

    def checkpoint(self, path):
        """Write :meth:`state_dict` to ``path`` (.npz).  A handle created with the same EngineConfig and restored from
        the file continues bit-identically."""
        np.savez(path, **self.state_dict())

    def restore(self, path):
        with np.load(path, allow_pickle=False) as z:
            self.load_state_dict({k: z[k] for k in z.files})

    # ------------------------------------------------------------------ measurement
    def profile(self, kernels=(N.KERNEL_ACTOR, N.KERNEL_SIM, N.KERNEL_CRITIC), stride=1, skip=0):
        """Bracket every ``stride``-th launch of the given kernels with HIP events on the engine's own
        stream (rcg_profile), the first one after ``skip`` (< stride) launches; ``False`` / empty stops recording."""
        mask = 0
        if kernels is True:
            mask = 7
        elif kernels:
            for k in kernels:
                mask |= 1 << int(k)
        if mask:
            stride = min(max(int(stride), 1), 0xfff)
            mask |= stride << 8
            mask |= (int(skip) % stride) << 20
        N.check(N.lib().rcg_profile(self._h, mask), self._h)

    def profile_pause(self):
        """Stop sampling without waiting for the launches still queued (RCG_PROFILE_PAUSE); the samples stay readable."""
        N.check(N.lib().rcg_profile(self._h, 0x80), self._h)

    def profile_read(self, kernel=N.KERNEL_ACTOR):
        """(total device ms, launches) of one kernel since ``profile(True)``."""
        ms, n = C.c_double(), C.c_int64()
        N.check(N.lib().rcg_profile_read(self._h, int(kernel), C.byref(ms), C.byref(n)), self._h)
        return ms.value, n.value

    def profile_samples(self, kernel=N.KERNEL_ACTOR) -> np.ndarray:
        """Duration (ms) of every sampled launch of one kernel since ``profile(...)``, in launch order
        (rcg_profile_samples): the dispatches' own start / end stamps."""
        n = C.c_int64()
        N.check(N.lib().rcg_profile_samples(self._h, int(kernel), None, 0, C.byref(n)), self._h)
        out = np.empty(max(n.value, 1), dtype=np.float64)
        N.check(N.lib().rcg_profile_samples(self._h, int(kernel), out.ctypes.data_as(C.POINTER(C.c_double)), n.value,
                                            C.byref(n)), self._h)
        return out[:n.value]

    def last_launch(self, kind=N.KERNEL_ACTOR):
        """Which kernel served the last launch of a kind (rcg_last_launch):
        ``{"kernel": "k_actor_dma", "kernel_id": 2, "variant": 0, "envs_per_wave": 8, "split": False}``."""
        kid, var, epw = C.c_int32(), C.c_int32(), C.c_int32()
        N.check(N.lib().rcg_last_launch(self._h, int(kind), C.byref(kid), C.byref(var), C.byref(epw)), self._h)
        return {"kernel": N.lib().rcg_kernel_name(kid.value).decode(), "kernel_id": kid.value, "variant": var.value & ~4096,
                "envs_per_wave": epw.value, "split": bool(var.value & 4096)}  # split: one half of a tick in two halves
