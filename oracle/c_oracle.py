"""ctypes wrapper of oracle/_build/liboracle.so (oracle/oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by rcognita_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import rcg_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "liboracle.so")


class OrcCfg(C.Structure):
    _fields_ = [("sys_id", C.c_int32), ("n_actor", C.c_int32), ("mode", C.c_int32), ("biquad", C.c_int32),
                ("critic_struct", C.c_int32), ("has_target", C.c_int32), ("clip", C.c_int32),
                ("substeps_per_tick", C.c_int32), ("gamma", C.c_double), ("h_pred", C.c_double),
                ("dt_sim", C.c_double), ("sampling_time", C.c_double), ("pars", C.c_double * 8),
                ("lo", C.c_double * 2), ("hi", C.c_double * 2), ("R1", C.c_double * 49), ("R2", C.c_double * 49),
                ("target", C.c_double * 8)]


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-Wall",
                           os.path.join(_HERE, "oracle.c"), "-o", LIB, "-lm"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(_HERE, "oracle.c")):
            build()
        _lib = C.CDLL(LIB)
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def to_c(cfg: O.OracleCfg) -> OrcCfg:
    c = OrcCfg()
    ds, du, n = cfg.ds, cfg.du, cfg.ds + cfg.du
    c.sys_id, c.n_actor, c.mode = cfg.sys_id, cfg.n_actor, cfg.mode
    c.biquad = 1 if cfg.stage_obj_struct == O.STAGE_BIQUADRATIC else 0
    c.critic_struct = cfg.critic_struct
    c.has_target = 0 if cfg.target is None else 1
    c.clip = 1 if (cfg.ctrl_bnds is not None and np.any(cfg.ctrl_bnds)) else 0
    c.substeps_per_tick = cfg.substeps_per_tick
    c.gamma, c.h_pred, c.dt_sim, c.sampling_time = cfg.gamma, cfg.pred_step_size, cfg.dt_sim, cfg.sampling_time
    for i, v in enumerate(np.asarray(cfg.pars).reshape(-1)[:8]):
        c.pars[i] = v
    for k in range(du):
        c.lo[k], c.hi[k] = cfg.ctrl_bnds[k]
    for i in range(n):
        for j in range(n):
            c.R1[i * n + j] = cfg.R1[i, j]
            c.R2[i * n + j] = 0.0 if cfg.R2 is None else cfg.R2[i, j]
    if cfg.target is not None:
        for i in range(ds):
            c.target[i] = cfg.target[i]
    return c


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _pars(cfg, pars, B):
    if pars is None:
        return np.zeros(8), 0
    p = np.zeros((B, 8))
    p[:, : pars.shape[1]] = pars
    return p, 1


def actor_cost(cfg, cand, obs, state_sys, pars=None, w=None, nthreads=1):
    """cand [B,K,N,du], obs/state_sys [B,ds] -> J [B,K]."""
    cand = np.ascontiguousarray(cand, dtype=np.float64)
    B, K = cand.shape[:2]
    obs = np.ascontiguousarray(np.broadcast_to(obs, (B, cfg.ds)), dtype=np.float64)
    xs = np.ascontiguousarray(np.broadcast_to(state_sys, (B, cfg.ds)), dtype=np.float64)
    pe, per = _pars(cfg, pars, B)
    cc = to_c(cfg)
    if not per:
        pe = np.ascontiguousarray(np.array(cc.pars[:8]))
    wa = None if w is None else np.ascontiguousarray(np.broadcast_to(w, (B, cfg.dc)), dtype=np.float64)
    J = np.empty((B, K))
    lib().orc_actor_cost_batch(C.byref(cc), B, K, _p(cand), _p(obs), _p(xs), _p(pe), per, _p(wa), cfg.dc, _p(J),
                               int(nthreads))
    return J


class CBatch:
    """Closed-loop state of B envs for orc_control_tick (AoS, float64)."""

    def __init__(self, cfg, state0, pars=None):
        self.cfg, self.cc = cfg, to_c(cfg)
        self.state = np.ascontiguousarray(np.array(state0, dtype=np.float64).reshape(-1, cfg.ds))
        B = self.B = self.state.shape[0]
        self.action = np.ascontiguousarray(np.broadcast_to(cfg.ctrl_bnds[:, 0] / 10.0, (B, cfg.du)).astype(np.float64))
        self.accum = np.zeros(B)
        self.step_idx = np.zeros(B, dtype=np.int32)
        self.best_J = np.zeros(B)
        self.best_idx = np.zeros(B, dtype=np.int32)
        self.pars, self.per = _pars(cfg, pars, B)
        if not self.per:
            self.pars = np.ascontiguousarray(np.array(self.cc.pars[:8]))
        self.w = np.ones((B, cfg.dc))

    def tick(self, cand, nthreads=1):
        cand = np.ascontiguousarray(cand, dtype=np.float64)
        K = cand.shape[1]
        lib().orc_control_tick(C.byref(self.cc), self.B, K, _p(cand), _p(self.state), _p(self.action), _p(self.accum),
                               _p(self.step_idx), _p(self.pars), self.per, _p(self.w), self.cfg.dc, _p(self.best_J),
                               _p(self.best_idx), int(nthreads))


def max_threads():
    return int(lib().orc_max_threads())
