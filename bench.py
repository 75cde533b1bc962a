#!/usr/bin/env python3
"""bench.py - env.control-steps/sec of the rcognita hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5] [--scaling weak|strong]

Workload, default (BASELINE.json configs[1], SURVEY.md 8d "C2"): Sys3WRobot, B = 65536 envs per GPU, RK4
dt = 0.01 (one substep per control tick), CtrlOptPred MPC, Nactor = 10, K = 256 candidate action
sequences per env streamed from HBM as a [B][K][N][du] tensor (the `_actor_cost(action_sqn, obs)`
operator shape) in FLOAT64 - the reference's own arithmetic width ("all reference arithmetic is float64"):
`value`, `ms_per_step`, `roofline` and `dtype` are the float64 tick; the float32 tick of the same workload (half the
candidate bytes) is measured in the same run, the same way (in-stream event bracket + dispatch stamps), and reported as
`value_f32` / `roofline_f32`.  One "step" = one env.control-step (unit U2) for every env of the batch:
rcg_control_tick = k_sim (closed_loop_rhs under RK4) + k_actor_dma (K rollouts + argmin + accum update).
Inputs are synthetic and resident in HBM before the timed region.
  --config C3   configs[2]: Sys2Tank, 131072 envs per GPU, Nactor = 20, RQL with the quadratic critic refitted every tick
                (k_critic_fit: env step + buffer push + TD fit in one launch), K = 256 candidates streamed (k_actor_dma, RQL
                instance) or `--regime generated`.
  --config C4   configs[3]: 524288 Sys3WRobot envs in total, sharded over the ranks (strong scaling unless
                --scaling weak), ONE all_gather of the per-env episode returns over RCCL in the timed region.
  --config C5   configs[4]: mixed pool 3wrobot + 3wrobot_NI + 2tank, 65536 envs per GPU sharded within each type,
                Nactor = 15, 256 generated candidates per env (VALU-bound regime).

Multi-GPU: one process per GPU.  `python bench.py --gpus N` from a plain shell starts the N ranks itself (the parent
never touches the GPU: it spawns N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits,
relays rank 0's JSON line and fails if any rank fails); under torchrun (`WORLD_SIZE` already set) the process IS a rank.

Timing: pre-spin (untimed, until the clock is steady) -> barrier + synchronize -> W warm-up steps -> [event] K steps
[event] -> K more steps without per-launch sampling (cross-check) -> the episode-end exchange, timed on its own ->
synchronize + barrier.  `ms_per_step` = the event pair's device time / K (max over ranks) + allgather_ms / 1000 (one
exchange per 1000-tick episode); nothing synchronises the host between the warm-up and the timed steps.

One JSON line on stdout (rank 0).  `roofline` prices the dominant kernel: algorithmic bytes per launch
/ its mean duration over >= 20 launches of the timed region, from start / stop events carried by the dispatches
themselves on the engine's own stream (mean, median, min reported; `kernel` is what rcg_last_launch says ran).  `parity` compares
outputs of the same run's kernels with the CPU oracle.  `cpu_baseline` is the C oracle (oracle/oracle.c, kind "port")
timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec; 6.29e12 measured copy)
VALU_PEAK = 7.9e13  # f32 lane-instructions/s: 256 CUs x 4 SIMDs x 32 lanes/clk... = 157.3 TFLOP/s FMA / 2
PRESPIN_S = 0.35    # untimed clock pre-spin: the first launches after the GPU wakes run 10-25 % slower
PRESPIN_MAX_S = 3.0
C4_TOTAL_ENVS = 524288
EPISODE_TICKS = 1000  # T = 1 000 control ticks per episode (10 s at the preset's t1); one exchange per episode


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=500)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--config", choices=["C2", "C3", "C4", "C5"], default="C2",
                   help="BASELINE.json configs[1] (default, the metric's config), configs[2], configs[3], configs[4]")
    p.add_argument("--scaling", choices=["weak", "strong"], default=None,
                   help="weak: --batch envs per GPU; strong: --batch envs in total (default: weak, C4: strong)")
    p.add_argument("--batch", type=int, default=None, help="envs per GPU (weak) or in total (strong)")
    p.add_argument("--candidates", type=int, default=256, help="K candidate sequences per env")
    p.add_argument("--nactor", type=int, default=None, help="horizon (default 10; C3: 20; C5: 15)")
    p.add_argument("--regime", choices=["streamed", "generated"], default=None)
    p.add_argument("--dtype", choices=["f32", "f64"], default=None,
                   help="element type of the headline tick.  Default f64 = the reference's own arithmetic width for the streamed "
                        "(HBM-bound) configs, with the f32 tick (half the candidate bytes) beside it as value_f32 / roofline_f32; "
                        "the generated-grid regime (--regime generated, --config C5) is a float32 regime (k_ticks_pk) and defaults to f32")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true")
    p.add_argument("--no-parity", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                   help="nccl = RCCL over xGMI (the real thing); gloo only to exercise the N>1 code path on one GPU")
    p.add_argument("--single-device", action="store_true",
                   help="debug: every rank uses cuda:0 (with --dist-backend gloo), to test the N>1 path on a 1-GPU box")
    p.add_argument("--profile-stride", type=int, default=-1,
                   help="time every n-th launch of the dominant kernel in the timed region (events carried by the "
                        "dispatch); -1 = chosen so that 20 .. 40 launches are sampled, 0 = none")
    p.add_argument("--tick-parts", type=int, default=None, choices=(0, 1, 2),
                   help="rcg_set_tick_parts of every handle.  Default: 2 for a single RQL / SQL handle (--config C3) - the explicit "
                        "opt-in to the pipelined split tick; this harness honours its contract (rcg_join in front of every event "
                        "it records on the handle's stream) - and 0 otherwise.  0 = the library's rule (since round 6: never on a "
                        "caller's stream, which is what this harness hands the handle; on a stream the handle owns an eligible tick of >= "
                        "65 536 envs runs as two halves on two internal streams), 1 = never, 2 = whenever eligible")
    p.add_argument("--parts", type=int, default=None,
                   help="handles per GPU, each on a stream of its own (rcognita_amd.pool.MixedPool(parts=...)): the critic fit "
                        "of one part runs under the actor kernel of another.  Default: 1 (since round 5 an RQL / SQL handle splits its own tick, --tick-parts)")
    p.add_argument("--force-dist", action="store_true",
                   help="initialise torch.distributed (RCCL with --dist-backend nccl) even at world_size 1 and run every "
                        "collective of the N>1 path: communicator creation, device all_gather, all_reduce, barrier")
    p.add_argument("--launch-timeout", type=float, default=3000.0, help="seconds the parent waits for its ranks")
    p.add_argument("--dry-launch", action="store_true",
                   help="ranks only rendezvous (gloo, CPU) and report their environment: tests the launcher without a GPU")
    a = p.parse_args(argv)
    if a.scaling is None:
        a.scaling = "strong" if a.config == "C4" else "weak"
    if a.batch is None:
        a.batch = C4_TOTAL_ENVS if (a.config == "C4" and a.scaling == "strong") else (131072 if a.config == "C3" else 65536)
    if a.nactor is None:
        a.nactor = {"C5": 15, "C3": 20}.get(a.config, 10)
    if a.regime is None:
        a.regime = "generated" if a.config == "C5" else "streamed"
    if a.dtype is None:
        a.dtype = "f32" if a.regime == "generated" else "f64"
    if a.parts is None:
        a.parts = 1  # (round 4: 2 for C3 - two handles on two streams hid the critic fit; since round 5 ONE handle does that itself)
    if a.parts > 1 and a.config == "C5":
        p.error("--parts applies to the single-system configs (C5 already runs one handle per system type)")
    if a.config == "C5" and a.regime != "generated":
        p.error("--config C5 is the generated-candidate workload (BASELINE configs[4]: 256-candidate grid search)")
    return a


# ---------------------------------------------------------------------------------------------------------------
# launcher: the parent of `python bench.py --gpus N` (never imports torch / librcg, never touches the GPU)
# ---------------------------------------------------------------------------------------------------------------
def refuse_dev_knobs():
    """Result- or schedule-changing developer knobs of librcg (rcg_sysops.hpp::DevKnobs) must not be
    set for a measurement: the bench line is quoted for the library as shipped."""
    bad = sorted(k for k in os.environ if k.startswith("RCG_"))
    if bad:
        raise SystemExit(f"bench.py: refusing to measure with developer knobs set in the environment: {', '.join(bad)} "
                         "(unset them; A/B experiments belong in tools/)")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(args, argv):
    """Start args.gpus ranks of this script (one process per GPU), wait, relay rank 0's stdout."""
    n = args.gpus
    port = int(os.environ.get("MASTER_PORT") or free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RCG_BENCH_LAUNCHER="self")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    deadline = time.monotonic() + args.launch_timeout
    outs = [None] * n
    failed = None
    pending = set(range(n))
    while pending and failed is None:
        for r in sorted(pending):
            try:
                outs[r] = procs[r].communicate(timeout=0.2)
                pending.discard(r)
                if procs[r].returncode != 0:
                    failed = r
                    break
            except subprocess.TimeoutExpired:
                pass
        if time.monotonic() > deadline:
            failed = -1
            break
    if failed is not None:  # one rank died or the job timed out: stop exactly the processes started above
        for r in pending:
            procs[r].kill()
        for r in pending:
            outs[r] = procs[r].communicate()
        who = "timeout" if failed < 0 else f"rank {failed} exited with {procs[failed].returncode}"
        sys.stderr.write(f"bench.py launcher: {who}\n")
        for r in range(n):
            if outs[r] and outs[r][1]:
                sys.stderr.write(f"--- rank {r} stderr (tail) ---\n{outs[r][1][-3000:]}\n")
        return 1
    for r in range(1, n):
        if outs[r][1]:
            sys.stderr.write(outs[r][1][-2000:])
    sys.stderr.write(outs[0][1][-4000:] if outs[0][1] else "")
    sys.stdout.write(outs[0][0])
    sys.stdout.flush()
    return 0


# ---------------------------------------------------------------------------------------------------------------
# workload pieces
# ---------------------------------------------------------------------------------------------------------------
C3_KW = dict(mode="RQL", critic_struct="quadratic", Ncritic=4, buffer_size=10, gamma=1.0)


def c2_engine_config(args, device, batch, dtype=None, env_id_base=0):
    import numpy as np

    from rcognita_amd import EngineConfig
    from rcognita_amd import _native as N

    if args.config == "C3":  # Sys2Tank with its preset's constants (presets/main_2tank.py:45-48, 199-211)
        from rcognita_amd.pool import preset_engine_config

        return preset_engine_config("2tank", batch, Nactor=args.nactor, dtype=dtype or args.dtype, device=device,
                                    env_id_base=env_id_base, **C3_KW), np.array([[0.0, 1.0]])

    bnds = np.array([[-300.0, 300.0], [-100.0, 100.0]])  # presets/main_3wrobot.py:207-211
    R1 = np.diag([1.0, 10.0, 1.0, 0, 0, 0, 0])  # presets/main_3wrobot.py R1_diag default
    return EngineConfig(sys_id=N.SYS_3WROBOT, batch=batch, dtype=dtype or args.dtype, device=device,
                        Nactor=args.nactor, mode="MPC", pars=[10.0, 1.0], ctrl_bnds=bnds, R1=R1, gamma=1.0,
                        dt_sim=0.01, sampling_time=0.01, pred_step_size=0.02, substeps_per_tick=1,
                        env_id_base=env_id_base), bnds


def c2_oracle_cfg(args):
    import numpy as np

    from oracle import rcg_oracle as O

    if args.config == "C3":
        return O.OracleCfg(sys_id=O.SYS_2TANK, n_actor=args.nactor, pars=[18.4, 24.4, 1.3, 1.0, 0.2],
                           ctrl_bnds=np.array([[0.0, 1.0]]), R1=np.diag([10.0, 10.0, 1.0]), target=[0.5, 0.5],
                           mode=O.MODE_RQL, critic_struct=O.CRITIC_QUADRATIC, n_critic=4, buffer_size=10, gamma=1.0,
                           dt_sim=0.1, sampling_time=0.1, pred_step_size=0.2)

    bnds = np.array([[-300.0, 300.0], [-100.0, 100.0]])
    return O.OracleCfg(sys_id=O.SYS_3WROBOT, n_actor=args.nactor, pars=[10.0, 1.0], ctrl_bnds=bnds,
                       R1=np.diag([1.0, 10.0, 1.0, 0, 0, 0, 0]), gamma=1.0, dt_sim=0.01, sampling_time=0.01,
                       pred_step_size=0.02)


def synth_state(seed, lo, hi, config="C2"):
    """x,y ~ U(-10,10), alpha ~ U(-pi,pi), v, omega ~ U(-1,1) (C3: h1 ~ U(0,2), h2 ~ U(-2,2)); the job's env
    g always gets the same state whatever the number of ranks (generated for the whole job, sliced to this rank's
    [lo, hi))."""
    import numpy as np

    rng = np.random.default_rng(seed)
    n = hi
    if config == "C3":
        return np.stack([rng.uniform(0, 2, n), rng.uniform(-2, 2, n)], axis=-1)[lo:hi]
    x = np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-np.pi, np.pi, n),
                  rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)], axis=-1)
    return x[lo:hi]


def actor_bytes_per_launch(B, K, N, du, ds, esz, streamed):
    """Algorithmic HBM bytes of one actor launch in tick mode."""
    per_env = ds * esz  # state read (obs == state_sys)
    per_env += du * esz + esz + 4  # action, best_J, best_idx writes
    per_env += 2 * esz + 2 * 4  # accum and step_idx read-modify-write
    if streamed:
        per_env += K * N * du * esz  # the candidate rows
    return B * per_env


def cpu_baseline(args, seconds):
    """C oracle (port of the same algorithm, f64) on the host cores, bounded sample of the workload."""
    import numpy as np

    from oracle import c_oracle as CO
    from oracle import rcg_oracle as O

    # the threads this job may really run: the cgroup / affinity share of the host (16 of 256 on a one-GPU box), NOT
    # omp_get_max_threads() - round 4 started 128 OpenMP threads on a 16-core share and labelled the result "128 cores"
    threads = max(1, min(usable_cores(), CO.max_threads()))
    cfg = c2_oracle_cfg(args)
    bnds = cfg.ctrl_bnds
    K = args.candidates
    Bc = 256 * threads
    rng = np.random.default_rng(99)
    cb = CO.CBatch(cfg, synth_state(1234, 0, Bc))
    if args.regime == "streamed":
        cand = bnds[:, 0] + (bnds[:, 1] - bnds[:, 0]) * rng.random((Bc, K, args.nactor, 2))
    else:
        cand = np.broadcast_to(O.grid_candidates(cfg, K)[None], (Bc, K, args.nactor, 2)).copy()
    t0 = time.perf_counter()
    cb.tick(cand, nthreads=threads)  # calibration tick (counted)
    one = time.perf_counter() - t0
    ticks = int(max(1, min(2000, (seconds - one) / max(one, 1e-6))))
    for _ in range(ticks):
        cb.tick(cand, nthreads=threads)
    dt = time.perf_counter() - t0
    n = Bc * (ticks + 1)
    return {"value": n / dt, "unit": "env-control-steps/s", "cores": threads, "threads": threads,
            "os_cpu_count": os.cpu_count(), "omp_max_threads": CO.max_threads(), "per_core": n / dt / threads, "kind": "port",
            "sample": f"{Bc} envs x {ticks + 1} ticks, K={K}, Nactor={args.nactor}, C oracle f64 + OpenMP on {threads} "
                      f"threads (= the cores this job owns), {dt:.1f} s"}


# the reference itself (imported from /root/reference in the build container, SciPy RK45 + SLSQP, one env, one core)
# ran 3.4 control steps/s at this horizon there; oracle/ref_loop.py - the same algorithm over the
# oracle's operators, pinned on the reference's F7 traces - ran 3.3 in the same container: calibration ratio 0.97
REF_IN_BUILD_CONTAINER = {"reference_steps_per_s_per_core": 3.4, "ref_loop_steps_per_s_per_core": 3.3}


def usable_cores():
    """Host cores this process may really use: os.cpu_count() capped by the affinity mask and the cgroup CPU quota (a GPU
    box hands a one-GPU job a share of the host's cores)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_reference_algorithm(args, seconds):
    """The reference ALGORITHM on the host cores, as BASELINE.md 4-1 prescribes it: one env per process - SciPy RK45 + SLSQP
    over the oracle's operators (oracle/ref_loop.py, pinned on traces captured from the reference) - and P such
    processes side by side, P = the cores this job may use; the aggregate and the per-core rate are both reported.
    Bounded by wall time.  The workers never touch the GPU (numpy + scipy only)."""
    P = usable_cores()
    env = {k: v for k, v in os.environ.items()}
    env.update(OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "oracle.ref_loop", "--seconds", str(seconds), "--nactor", str(args.nactor)]
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for _ in range(P)]
    res = []
    for pr in procs:
        out, _ = pr.communicate(timeout=seconds * 4 + 120)
        lines = [l for l in out.splitlines() if l.startswith("{")]
        if pr.returncode == 0 and lines:
            res.append(json.loads(lines[-1]))
    wall = time.perf_counter() - t0
    if not res:
        raise ImportError("no reference-algorithm worker finished (SciPy missing on this box?)")
    rates = [r["ticks"] / r["seconds"] for r in res]
    cal = REF_IN_BUILD_CONTAINER
    return {"value": float(sum(rates)), "unit": "env-control-steps/s", "cores": len(res),
            "per_core": float(sum(rates) / len(rates)), "os_cpu_count": os.cpu_count(),
            "kind": "reference algorithm (SciPy RK45 + SLSQP over the oracle's operators, preset main_3wrobot), one env "
                    "per process",
            "sample": f"{len(res)} processes x 1 env, {sum(r['ticks'] for r in res)} control ticks, "
                      f"{sum(r['nfev_actor'] for r in res)} _actor_cost evaluations, {seconds:.0f} s each ({wall:.1f} s wall "
                      "with interpreter start-up)",
            "calibration": {**cal, "ref_loop_over_reference": cal["ref_loop_steps_per_s_per_core"] /
                            cal["reference_steps_per_s_per_core"],
                            "note": "measured in the build container, where the reference can be imported; it cannot "
                                    "travel to this box"}}


def parity_check_pool(args, device, stream_ptr, K, tol=1e-5, n_env=192, ticks=2):
    """--config C5: the mixed pool's kernels (generated grid, one handle per system type) on fresh small handles of each
    type, tick by tick against the CPU oracle (oracle/parity.py), same rule as parity_check."""
    import numpy as np

    from oracle import parity as PAR
    from oracle import rcg_oracle as O
    from rcognita_amd import Engine
    from rcognita_amd import _native as N
    from rcognita_amd.pool import PRESETS, preset_engine_config

    rep = PAR.TickReport()
    tol = tol if args.dtype == "f32" else 1e-11
    real = np.float32 if args.dtype == "f32" else np.float64
    out = {"ok": True, "tol": tol, "types": {}}
    rng = np.random.default_rng(77)
    for name, p in PRESETS.items():
        eng = Engine(preset_engine_config(name, n_env, Nactor=args.nactor, dtype=args.dtype, device=device))
        eng.set_stream(stream_ptr)
        x0 = pool_states(rng, name, n_env)
        eng.set_state(x0)
        ocfg = O.OracleCfg(sys_id=p["sys_id"], n_actor=args.nactor, pars=p["pars"], ctrl_bnds=np.array(p["ctrl_bnds"]),
                           R1=np.diag(np.array(p["R1"], dtype=float)), target=p["target"], dt_sim=p["dt"],
                           sampling_time=p["dt"], pred_step_size=p["dt"] * p["mult"])
        env = O.new_batch(ocfg, x0.astype(real).astype(np.float64))
        cand_host = O.grid_candidates(ocfg, K)
        try:
            for t in range(ticks):
                eng.control_tick(None, K=K)
                env = PAR.check_tick(ocfg, env, cand_host, PAR.device_fields(eng, N, critic=False), tol=tol, report=rep,
                                     what=f"bench parity {name} tick {t}")
            out["types"][name] = {"ok": True, "kernel": eng.last_launch(N.KERNEL_ACTOR)["kernel"]}
        except AssertionError as e:
            out["ok"] = False
            out["types"][name] = {"ok": False, "error": str(e)[:300]}
        eng.close()
    out.update(rep.as_dict())
    return out


def parity_check(args, device, stream_ptr, x0, cand, K, tol=1e-5, n_sample=64, ticks=2):
    """A fresh run of the SAME kernels on the SAME inputs (this rank's states and candidate tensor), `ticks` control
    ticks, compared tick by tick with the CPU oracle on a sample of envs (oracle/parity.py: tie-aware best_idx, state /
    action / best_J / accum within `tol`, int32 step counters exact).  f32 tolerance: the north star's 1e-5."""
    import numpy as np

    from oracle import parity as PAR
    from oracle import rcg_oracle as O
    from rcognita_amd import Engine
    from rcognita_amd import _native as N

    B = x0.shape[0]
    ecfg, _ = c2_engine_config(args, device, B)
    eng = Engine(ecfg)
    eng.set_stream(stream_ptr)
    eng.set_state(x0)
    ocfg = c2_oracle_cfg(args)
    sel = np.sort(np.random.default_rng(7).choice(B, min(n_sample, B), replace=False))
    real = np.float32 if args.dtype == "f32" else np.float64
    env = O.new_batch(ocfg, x0[sel].astype(real).astype(np.float64))
    if cand is not None:
        import torch

        cand_host = cand[torch.as_tensor(sel, device=cand.device)].cpu().numpy().astype(np.float64)
    else:
        cand_host = O.grid_candidates(ocfg, K)
    rep = PAR.TickReport()
    tol = tol if args.dtype == "f32" else 1e-11
    critic = args.config == "C3"
    if critic:
        ticks = 14  # past the point where the buffers (10 rows) have filled: the TD fits are non-trivial from tick 8 on
    # critic weights solve a least-squares problem regularised at 1e-8 of its scale: float64 at 1e-6 per tick
    over = {"w_critic": 1e-6, "best_J": 1e-7} if (critic and args.dtype == "f64") else None  # (as tests/test_hip_critic.py)
    try:
        for t in range(ticks):
            eng.control_tick(cand, K=K)
            dev = {k: v[sel] for k, v in PAR.device_fields(eng, N, critic=critic).items()}
            env = PAR.check_tick(ocfg, env, cand_host, dev, tol=tol, report=rep, what=f"bench parity tick {t}",
                                 tol_over=over)
        out = {"ok": True, "tol": tol, **rep.as_dict()}
    except AssertionError as e:
        out = {"ok": False, "tol": tol, "error": str(e)[:500], **rep.as_dict()}
    eng.close()
    return out


# ---------------------------------------------------------------------------------------------------------------
_STDOUT_KEEP = None


def stdout_to_stderr():
    """The contract is ONE JSON line on stdout, and libraries greet there (RCCL 2.26 prints five lines of versions when its first
    communicator comes up; gloo announces its peers): while a process group exists, file descriptor 1 points at stderr."""
    global _STDOUT_KEEP
    if _STDOUT_KEEP is None:
        sys.stdout.flush()
        _STDOUT_KEEP = os.dup(1)
        os.dup2(2, 1)


def stdout_back():
    """Before the JSON line: flush what Python and the C library still hold for descriptor 1, then point it at stdout again."""
    global _STDOUT_KEEP
    if _STDOUT_KEEP is not None:
        sys.stdout.flush()
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(_STDOUT_KEEP, 1)
        os.close(_STDOUT_KEEP)
        _STDOUT_KEEP = None


def dry_rank(args, rank, local_rank, world):
    """--dry-launch: rendezvous over gloo on the CPU and report what the launcher handed to each rank."""
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    stdout_to_stderr()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    mine = torch.tensor([rank, local_rank, os.getpid()], dtype=torch.int64)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    from rcognita_amd.parallel import shard_range

    lo, hi = shard_range(args.batch if args.scaling == "strong" else args.batch * world, rank, world)
    span = torch.tensor([lo, hi], dtype=torch.int64)
    spans = [torch.empty_like(span) for _ in range(world)]
    dist.all_gather(spans, span)
    stdout_back()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "rccl_ranks": dist.get_world_size(),
                          "launcher": os.environ.get("RCG_BENCH_LAUNCHER", "external"),
                          "ranks": [{"rank": int(p[0]), "local_rank": int(p[1]), "pid": int(p[2]),
                                     "envs": [int(s[0]), int(s[1])]} for p, s in zip(parts, spans)],
                          "scaling": args.scaling, "config": args.config}))
    dist.destroy_process_group()


def pool_states(rng, name, n):
    """Synthetic initial states of the mixed pool, per system type."""
    import numpy as np

    if name == "3wrobot":
        return np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-np.pi, np.pi, n),
                         rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)], axis=-1)
    if name == "3wrobotNI":
        return np.stack([rng.uniform(-10, 10, n), rng.uniform(-10, 10, n), rng.uniform(-np.pi, np.pi, n)], axis=-1)
    return np.stack([rng.uniform(0, 2, n), rng.uniform(-2, 2, n)], axis=-1)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    launcher = os.environ.pop("RCG_BENCH_LAUNCHER", None)  # set by launch() for its children only; not a library knob
    refuse_dev_knobs()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher and must not initialise the GPU
        sys.exit(launch(args, argv))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        args.gpus = world
    launcher = ("bench.py self-spawn" if launcher else ("torchrun/external" if world > 1 else "single process"))
    if args.dry_launch:
        os.environ["RCG_BENCH_LAUNCHER"] = launcher
        return dry_rank(args, rank, local_rank, world)

    import numpy as np
    import torch  # device memory for the synthetic candidates, streams, events, torch.distributed (plumbing)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    if args.single_device:
        if args.dist_backend == "nccl" and world > 1:
            raise SystemExit("bench.py: --single-device puts every rank on cuda:0, which RCCL refuses (one communicator "
                             "rank per GPU); use it with --dist-backend gloo")
        local_rank = 0
    if local_rank >= torch.cuda.device_count():  # never alias two ranks onto one GPU silently
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but only {torch.cuda.device_count()} "
                         "device(s) are visible (to exercise the N>1 path on one GPU: --single-device --dist-backend gloo)")
    torch.cuda.set_device(local_rank)
    dist = None
    coll_dev = torch.device("cuda", local_rank)  # where collective payloads live
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
        stdout_to_stderr()
        if args.dist_backend == "nccl":  # RCCL: the communicator is bound to this rank's GPU at creation
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=coll_dev)
        else:
            coll_dev = torch.device("cpu")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from rcognita_amd import Engine
    from rcognita_amd import _native as N
    from rcognita_amd.parallel import gather_summaries, merge_summaries, shard_range

    K, Nh = args.candidates, args.nactor
    total_envs = args.batch if args.scaling == "strong" else args.batch * world
    main_stream = torch.cuda.current_stream()
    stream_ptr = main_stream.cuda_stream
    tdtype = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    streamed = args.regime == "streamed"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- this rank's shard of the job -----------------------------------------------------------------------------
    cand, x0, pool = None, None, None
    pool_cand = None
    if args.config == "C5" or args.parts > 1:
        from rcognita_amd.pool import MixedPool

        if args.config == "C5":
            counts = {"3wrobot": total_envs // 3 + total_envs % 3, "3wrobotNI": total_envs // 3, "2tank": total_envs // 3}
            kw = {}
        else:  # one system type cut into --parts handles: the job's envs [lo, hi) of this rank
            sysname = "2tank" if args.config == "C3" else "3wrobot"
            counts = {sysname: total_envs}
            kw = dict(parts=args.parts, **(C3_KW if args.config == "C3" else dict(mode="MPC")))
        # the handles are independent: each runs on a HIP stream of its own, so the tail of one kernel overlaps the head of
        # the next and (RQL / SQL) the latency-bound critic fit of one part runs under the HBM-bound actor kernel of another.
        # The streams are torch's here, so that this harness can record its events ON them.
        pool = MixedPool(counts, rank=rank, world=world, device=local_rank, dtype=args.dtype, Nactor=Nh, own_streams=False,
                         **kw)
        streams = [torch.cuda.Stream() for _ in pool.segments]
        pool.set_streams([st.cuda_stream for st in streams])
        engines = [s.engine for s in pool.segments]
        B = pool.n_envs
        if args.config == "C5":
            rng = np.random.default_rng(1234 + rank)
            pool.set_states({s.name: pool_states(rng, s.name, s.hi - s.lo) for s in pool.segments})
            tick = lambda: pool.control_tick(K)
        else:
            lo, hi = shard_range(total_envs, rank, world)
            x0 = synth_state(1234, lo, hi, args.config)
            pool.set_states({sysname: x0})
            _, bnds = c2_engine_config(args, local_rank, 1)
            if streamed:
                g = torch.Generator(device="cuda")
                g.manual_seed(1234 + rank)
                blo = torch.tensor(bnds[:, 0], device="cuda", dtype=tdtype)
                bhi = torch.tensor(bnds[:, 1], device="cuda", dtype=tdtype)
                cand = (torch.rand((B, K, Nh, engines[0].du), generator=g, device="cuda", dtype=tdtype) * (bhi - blo)
                        + blo).contiguous()
                torch.cuda.synchronize()  # written on torch's current stream, read on the handles' streams
                pool_cand = {sysname: cand}
            tick = lambda: pool.control_tick(K, pool_cand, ordered=True)
    else:
        lo, hi = shard_range(total_envs, rank, world)
        B = hi - lo
        ecfg, bnds = c2_engine_config(args, local_rank, B, env_id_base=lo)
        eng = Engine(ecfg)
        eng.set_stream(stream_ptr)
        x0 = synth_state(1234, lo, hi, args.config)
        eng.set_state(x0)
        if streamed:
            g = torch.Generator(device="cuda")
            g.manual_seed(1234 + rank)
            blo = torch.tensor(bnds[:, 0], device="cuda", dtype=tdtype)
            bhi = torch.tensor(bnds[:, 1], device="cuda", dtype=tdtype)
            cand = (torch.rand((B, K, Nh, eng.du), generator=g, device="cuda", dtype=tdtype) * (bhi - blo) + blo).contiguous()
        engines = [eng]
        streams = [main_stream]
        tick = lambda: eng.control_tick(cand, K=K)
    du, ds = (1, 2) if args.config == "C3" else (2, 5)

    if args.tick_parts is None:
        args.tick_parts = 2 if (args.config == "C3" and len(engines) == 1) else 0
    for e in engines:  # (several handles already overlap each other: no split inside them unless asked for)
        e.set_tick_parts(args.tick_parts if (args.tick_parts or len(engines) == 1) else 1)

    def record_all():
        """One timing event per engine stream, recorded now (in-stream, no host wait).  A handle that splits its tick over two
        internal streams first orders its own stream behind them (rcg_join: two event waits, no host wait)."""
        for e in engines:
            e.join()
        evs = []
        for st in streams:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(st)
            evs.append(ev)
        return evs

    def align_streams():
        """Several handles on streams of their own drift apart (the stream with the longest kernels carries a backlog):
        make every stream wait, ON THE DEVICE, until all of them have arrived here, so that a timed region starts for all
        handles at the same instant.  No host wait; a no-op for a single stream."""
        if len(streams) < 2:
            return
        here = record_all()
        for i, st in enumerate(streams):
            for j, ev in enumerate(here):
                if j != i:
                    st.wait_event(ev)

    def span_ms(starts, stops):
        """Device time from the (aligned) start to the latest stop over the engines' streams."""
        return max(a.elapsed_time(b) for a in starts for b in stops)

    def chunk_ms_of(starts, stops):
        """Pre-spin chunks are not aligned: each stream's own time for the chunk, the slowest stream's counts."""
        return max(a.elapsed_time(b) for a, b in zip(starts, stops))

    # ---- everything the timed region needs exists BEFORE the clock pre-spin: no allocation, no event creation, no host
    # synchronisation stands between the warm-up steps and the timed steps ----------------------------------------------
    Bmax = B
    if dist is not None:  # ragged shards: the all_gather payload is padded to the largest shard
        bm = torch.tensor([B], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(bm, op=dist.ReduceOp.MAX)
        Bmax = int(bm.item())
    returns_dev = torch.zeros(Bmax, device="cuda", dtype=tdtype)
    gathered = [torch.empty(Bmax, device=coll_dev, dtype=tdtype) for _ in range(world)] if dist is not None else None
    torch.cuda.synchronize()  # the buffers above were filled on torch's stream; the engines may run on others

    def exchange_returns():
        """Episode-end exchange: ONE all_gather of the per-env running returns over RCCL."""
        off = 0
        for e in engines:
            N.check(N.lib().rcg_get_field(e._h, N.FIELD_ACCUM, returns_dev[off:].data_ptr(), N.DEVICE), e._h)
            off += e.B
        for st in streams:  # the copies ran on the engines' streams: the collective (torch's stream) comes after them
            if st is not main_stream:
                main_stream.wait_stream(st)
        dist.all_gather(gathered, returns_dev if coll_dev.type == "cuda" else returns_dev.cpu())

    if dist is not None:
        exchange_returns()  # untimed: RCCL builds its rings / channels on the first collective of each kind
        torch.cuda.synchronize()

    # ---- untimed: clock pre-spin (>= PRESPIN_S of the same kernels) ----------------------------------------------------
    # The GPU reaches its steady clock after ~100 ticks (20 ms) of CONTINUOUS work and falls back within 10 ms of idling
    # (tools/clock_ramp.py), so the spin never lets the queue run dry: the host waits on the events recorded two chunks
    # ago (on the engines' own streams).
    prespin = 0
    t_spin = time.perf_counter()
    evs, chunk_ms = [], []
    while True:
        for _ in range(32):
            tick()
        prespin += 32
        evs.append(record_all())
        if len(evs) >= 4:
            for ev in evs[-3]:
                ev.synchronize()  # two chunks stay queued behind it: the GPU never runs dry
            chunk_ms.append(chunk_ms_of(evs[-4], evs[-3]))
            spun = time.perf_counter() - t_spin
            # steady = the last four chunks within 1.5 % of each other (a box that has just been handed over can need
            # longer than a warm one); never less than PRESPIN_S, never more than PRESPIN_MAX_S
            last = chunk_ms[-4:]
            steady = len(last) == 4 and (max(last) - min(last)) <= 0.015 * min(last)
            if (spun >= PRESPIN_S and steady) or spun >= PRESPIN_MAX_S:
                break

    # ---- the contract's bracket: barrier + synchronize, W untimed warm-up steps, then EXACTLY K timed steps between two
    # in-stream events (no host synchronisation in between: the timed steps run in the flow the warm-up steps started),
    # then synchronize + barrier.  Per-launch durations of the dominant kernel(s) come from events carried by the
    # dispatches themselves (rcg_profile: hipExtLaunchKernelGGL start / stop stamps), at least 20 of them.
    prof_kernels = (N.KERNEL_ACTOR, N.KERNEL_CRITIC) if args.config == "C3" else (N.KERNEL_ACTOR,)
    stride = 0
    if args.profile_stride != 0:
        stride = args.profile_stride if args.profile_stride > 0 else max(1, args.steps // 32)
        if args.steps // stride < 20:
            stride = max(1, args.steps // 20)
    barrier()
    t_wall0 = time.perf_counter()
    for _ in range(args.warmup):
        tick()
    if stride:
        for e in engines:
            e.profile(prof_kernels, stride=stride)  # host-side switch only: nothing is pending, nothing is waited for
    align_streams()
    ev_start = record_all()
    for _ in range(args.steps):
        tick()
    ev_stop = record_all()
    for e in engines:
        e.profile_pause()
    # the same K steps again without any per-launch event: what the sampling itself costs the stream
    align_streams()
    ev_start2 = record_all()
    for _ in range(args.steps):
        tick()
    ev_stop2 = record_all()
    # the episode-end exchange, timed on its own: the design makes ONE exchange per episode (EPISODE_TICKS control ticks),
    # so its cost enters a step as allgather_ms / EPISODE_TICKS - not once per K steps of a short run
    n_exch = 5 if dist is not None else 0
    ex_a, ex_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if n_exch:
        for st in streams:
            if st is not main_stream:
                main_stream.wait_stream(st)
        ex_a.record(main_stream)
        for _ in range(n_exch):
            exchange_returns()
        ex_b.record(main_stream)
    barrier()
    wall_s = time.perf_counter() - t_wall0
    compute_ms = span_ms(ev_start, ev_stop)
    unbracketed_ms = span_ms(ev_start2, ev_stop2)
    allgather_ms = (ex_a.elapsed_time(ex_b) / n_exch) if n_exch else 0.0
    if dist is not None:  # MAX over ranks (device times of each rank's own stream)
        tmax = torch.tensor([compute_ms, unbracketed_ms, allgather_ms], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        compute_ms, unbracketed_ms, allgather_ms = (float(v) for v in tmax.tolist())
    step_ms_compute = compute_ms / args.steps
    step_ms = step_ms_compute + allgather_ms / EPISODE_TICKS
    dt = step_ms * args.steps * 1e-3

    samples = {k: np.concatenate([e.profile_samples(k) for e in engines]) for k in prof_kernels} if stride else {}
    for e in engines:
        e.profile(False)
    act = samples.get(N.KERNEL_ACTOR, np.zeros(0))
    actor_n = int(act.size)
    actor_ms = float(act.sum())
    launch_info = [e.last_launch(N.KERNEL_ACTOR) for e in engines]

    summ = merge_summaries([e.episode_stats(from_accum=True)[0] for e in engines])
    total = gather_summaries(summ, dist, device=coll_dev, force=args.force_dist)  # per-shard summaries -> whole job
    ticks_done = prespin + args.warmup + 2 * args.steps
    for e in engines:
        steps_idx = e.get_field(N.FIELD_STEP_IDX)
        assert int(steps_idx.min()) == int(steps_idx.max()) == ticks_done, "step counter mismatch"
    rank_info = None
    if dist is not None:
        mine = torch.tensor([rank, local_rank, B], dtype=torch.int64, device=coll_dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        rank_info = [{"rank": int(p[0]), "device": int(p[1]), "envs": int(p[2])} for p in parts]
        if coll_dev.type == "cuda":  # the gathered returns really are everyone's: compare their sum with the summaries
            allret = torch.stack(gathered).double()
            got = float(sum(allret[r, :ri["envs"]].sum().item() for r, ri in enumerate(rank_info)))
            assert abs(got - total["sum"]) <= 1e-6 * max(abs(total["sum"]), 1.0), (got, total["sum"])

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    units = total_envs * args.steps
    value = units / dt
    if args.config == "C5":
        bytes_launch = sum(actor_bytes_per_launch(e.B, K, Nh, e.du, e.ds, esz, False) for e in engines)
    else:  # one launch = one handle's envs (--parts handles of equal size per tick)
        Bl = engines[0].B
        bytes_launch = actor_bytes_per_launch(Bl, K, Nh, du, ds, esz, streamed)
        if args.config == "C3":
            bytes_launch += Bl * engines[0].dc * esz  # the env's critic weights travel with its state
    actor_avg_s = (actor_ms / max(actor_n, 1)) * 1e-3
    if args.config == "C5":
        # one launch per segment, on streams of their own: the launches overlap, so the per-tick figure is the device time of
        # a tick where that is shorter than the sum of the three kernel times
        actor_avg_s = min(actor_avg_s * len(engines), step_ms_compute * 1e-3)
    achieved = bytes_launch / actor_avg_s if actor_avg_s > 0 else 0.0
    per_launch = None
    split_inside = bool(launch_info and launch_info[0].get("split"))  # the handle ran its tick as two halves (rcg_set_tick_parts)
    if split_inside and args.config != "C5" and len(engines) == 1:
        # two half-batch launches per tick on two internal streams, overlapping each other and the other half's fit: as with
        # --parts handles, the roofline is taken at tick level
        per_launch = {"algorithmic_bytes": bytes_launch / 2, "avg_launch_ms": actor_avg_s * 1e3,
                      "frac_of_one_overlapped_launch": (bytes_launch / 2) / actor_avg_s / HBM_PEAK if actor_avg_s > 0 else None,
                      "note": "the tick runs as two half-batch launches on two internal streams (rcg_set_tick_parts): they "
                              "overlap, so a launch's own duration is not a statement about the memory system"}
        actor_avg_s = step_ms_compute * 1e-3
        achieved = bytes_launch / actor_avg_s
    if args.config != "C5" and len(engines) > 1:
        # --parts handles on streams of their own: their launches OVERLAP on the GPU, so a launch's own duration (what the
        # dispatch stamps and a rocprofv3 trace show) is stretched by its neighbour and bytes / duration of ONE launch says
        # nothing about the memory system.  The roofline is then taken at tick level: the algorithmic bytes of ALL the
        # tick's actor launches over the device time of a tick (which also contains the env steps / critic fits).
        per_launch = {"algorithmic_bytes": bytes_launch, "avg_launch_ms": actor_avg_s * 1e3,
                      "frac_of_one_overlapped_launch": achieved / HBM_PEAK,
                      "note": "launches of the handles overlap: not a statement about the memory system"}
        bytes_launch = bytes_launch * len(engines)
        actor_avg_s = step_ms_compute * 1e-3
        achieved = bytes_launch / actor_avg_s
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and args.config != "C5":
        try:
            mode_tag = "_RQL" if args.config == "C3" else ""
            sysname = "2tank" if args.config == "C3" else "3wrobot"
            Bl = engines[0].B  # envs of ONE launch
            keys = [f"k_actor_{args.regime}_{sysname}{mode_tag}_B{Bl}_K{K}_N{Nh}_{args.dtype}"]
            if args.config != "C3":
                keys.append(f"k_actor_{args.regime}_B{Bl}_K{K}_N{Nh}_{args.dtype}")  # rounds 1-2 key of the C2 shapes
            table = json.load(open(pmc))
            for key in keys:
                if key in table:
                    traffic = table[key].get("hbm_bytes_per_launch")
                    traffic_src = (f"profiles/pmc_traffic.json[{key}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                   "this command, stored; not re-measured in this run)")
                    break
        except Exception:
            traffic = None
    if traffic is not None and per_launch is not None and len(engines) > 1:
        traffic *= len(engines)  # tick level, as `achieved`
    if traffic is not None and split_inside and len(engines) == 1:
        traffic *= 2  # the stored counter pass saw the tick's two half-batch launches one by one: tick level, as `achieved`
    kinfo = launch_info[0]
    width = "float32 storage and arithmetic (SURVEY 8a-1 / 8d; the reference computes in float64: run without --dtype f32)" \
        if args.dtype == "f32" else "float64, the reference's width (float32 beside it: value_f32 / roofline_f32)"
    workload = {
        "C2": f"Sys3WRobot B={args.batch}/GPU RK4 dt=0.01 S=1, CtrlOptPred MPC Nactor={Nh}, K={K} {args.regime} "
              f"candidates (BASELINE configs[1]); headline in {width}",
        "C3": f"Sys2Tank B={args.batch}/GPU RK4 dt=0.1 S=1, CtrlOptPred RQL Nactor={Nh} + quadratic critic TD fit every tick "
              f"(Ncritic=4, buffer 10), K={K} {args.regime} candidates (BASELINE configs[2])",
        "C4": f"Sys3WRobot {total_envs} envs sharded over {world} rank(s), RK4 dt=0.01, MPC Nactor={Nh}, K={K} "
              f"{args.regime} candidates, all_gather of episode returns (BASELINE configs[3])",
        "C5": f"mixed pool 3wrobot+3wrobot_NI+2tank, {total_envs} envs over {world} rank(s) sharded within each type, "
              f"Nactor={Nh}, K={K} generated grid (BASELINE configs[4])",
    }[args.config]
    metric = "env-control-steps/sec (whole node), 3wrobot Nactor=10"
    if args.config != "C2" or Nh != 10:
        metric = f"env-control-steps/sec (whole node), {args.config} Nactor={Nh}"
    out = {
        "metric": metric,
        "value": value,
        "unit": "env-control-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_ms,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "value_definition": "total envs x K timed steps / (device time of the K steps, max over ranks, + K x allgather_ms / "
                            f"{EPISODE_TICKS}): the design makes ONE exchange per {EPISODE_TICKS}-tick episode (SURVEY 8d, 8e); "
                            "rounds 1-2 charged one exchange to the K steps and used the host clock between barriers - that "
                            "figure is timing.value_k_steps_plus_one_exchange",
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": workload, "config": args.config, "envs_total": total_envs, "envs_rank0": B,
                   "handles_per_gpu": len(engines), "envs_per_handle": [e.B for e in engines],
                   "candidates": K, "nactor": Nh, "regime": args.regime, "parallelism": f"env-shard x{world}",
                   "actor_cost_evals_per_s": value * K, "prespin_ticks_untimed": prespin,
                   "prespin_last_chunks_ms_per_tick": [round(c / 32, 5) for c in chunk_ms[-4:]]},
        # how `value` was timed: K steps between two in-stream events (max over ranks), no host synchronisation between
        # the W warm-up steps and the K timed steps; the episode-end exchange enters per episode, not per K steps
        "timing": {"clock": "HIP events recorded in-stream around the K timed steps (device time, max over ranks)",
                   "ms_per_step_compute": step_ms_compute,
                   "value_compute_only": total_envs / (step_ms_compute * 1e-3),
                   "value_k_steps_plus_one_exchange": units / ((compute_ms + allgather_ms) * 1e-3),
                   "allgather_ms": allgather_ms if dist is not None else None,
                   "episode_ticks": EPISODE_TICKS,
                   "allgather_ms_per_step": (allgather_ms / EPISODE_TICKS) if dist is not None else None,
                   "unbracketed_ms_per_step": unbracketed_ms / args.steps,
                   "wall_ms_per_step_incl_warmup_and_crosscheck": wall_s * 1e3 / (args.warmup + 2 * args.steps),
                   "note": "unbracketed = the same K steps repeated right behind the timed ones with no per-launch event "
                           "(the sampling costs the stream ~7 us per sampled launch); wall = host clock from the barrier "
                           "in front of the warm-up to the barrier behind everything, over all W + 2K steps (+ exchanges)"},
        "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
        "dist_backend": (args.dist_backend if dist is not None else None),
        "launcher": launcher,
        "ranks": rank_info,
        "roofline": {"bound": "hbm", "kernel": kinfo["kernel"], "kernel_variant": kinfo["variant"],
                     "envs_per_wave": kinfo["envs_per_wave"],
                     "kernel_source": "rcg_last_launch (the library's own dispatch)",
                     "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": bytes_launch, "avg_launch_ms": actor_avg_s * 1e3,
                     "median_launch_ms": float(np.median(act)) if actor_n else None,
                     "min_launch_ms": float(act.min()) if actor_n else None,
                     "max_launch_ms": float(act.max()) if actor_n else None,
                     "frac_at_median": (bytes_launch / (float(np.median(act)) * 1e-3) / HBM_PEAK) if actor_n and
                     args.config != "C5" else None,
                     "launches_timed": actor_n, "event_stride": stride, "overlapped_handles": per_launch,
                     "event_kind": "start / stop events carried by the dispatch (hipExtLaunchKernelGGL), timed region only",
                     "note": ("streamed regime: HBM-bound; the candidate tensor is STATIC - the same caller-owned tensor is "
                              "re-read from HBM every tick (1.3 GB at C2, 5 x the Infinity Cache): this is SURVEY 8d's "
                              "operator shape, _actor_cost(action_sqn, ...) over given sequences, not a closed loop whose "
                              "candidates change per tick - for that see value_closed_loop" if streamed else
                              "generated regime is VALU-bound (see secondary.generated_grid.roofline_valu); the HBM "
                              "fraction is reported for completeness only")},
        "returns_summary": total,
    }

    if N.KERNEL_CRITIC in samples and samples[N.KERNEL_CRITIC].size:
        cs = samples[N.KERNEL_CRITIC]
        out["roofline"]["critic_fit_kernel"] = {"kernel": engines[0].last_launch(N.KERNEL_CRITIC)["kernel"],
                                                "avg_launch_ms": float(cs.mean()), "median_launch_ms": float(np.median(cs)),
                                                "min_launch_ms": float(cs.min()), "launches_timed": int(cs.size),
                                                "bound": "latency of the longest active-set walk (DESIGN.md 4)"}
    if not streamed:
        # the generated-candidate regime is bound by VALU instruction issue, not by HBM: price it in lane-instructions/s
        #  with the instruction counts of the stored SQ_INSTS_VALU pass
        lane_instr, missing = 0.0, False
        for e in engines:
            name = {N.SYS_3WROBOT: "3wrobot", N.SYS_3WROBOT_NI: "3wrobotNI", N.SYS_2TANK: "2tank"}[int(e.cfg.sys_id)]
            key = (f"k_actor_generated_{name}_N{Nh}_{args.dtype}_C5" if args.config == "C5"
                   else (f"k_actor_generated_{name}_N{Nh}_RQL_{args.dtype}" if args.config == "C3"
                         else f"k_actor_generated_{name}_N{Nh}_{args.dtype}"))
            ipe = valu_instr_per_eval(key)
            if not ipe:
                missing = True
                break
            lane_instr += e.B * K * ipe["valu_instr_per_eval"]
            if args.config != "C5":  # equal handles, one launch each per tick: price ONE launch
                break
        hbm = dict(out["roofline"])
        if missing:  # no stored instruction count for this shape: say so instead of pricing a VALU-bound kernel in bytes
            out["roofline"].update(bound="valu", achieved=None, frac=None, unit="lane-instr/s", peak=VALU_PEAK,
                                   note="generated regime is VALU-bound; no SQ_INSTS_VALU pass is stored for this shape "
                                        "(tools/valu_probe.py + tools/prof_summary.py --valu)",
                                   hbm_frac_for_completeness=hbm["frac"])
        if not missing and actor_avg_s > 0:
            out["roofline"] = {"bound": "valu", "kernel": "k_actor (generated level grid)",
                               "achieved": lane_instr / actor_avg_s, "peak": VALU_PEAK, "unit": "lane-instr/s",
                               "frac": lane_instr / actor_avg_s / VALU_PEAK, "traffic": None,
                               "lane_instructions_per_tick": lane_instr, "avg_actor_ms_per_tick": actor_avg_s * 1e3,
                               "source": "profiles/valu_instr.json (rocprofv3 --pmc SQ_INSTS_VALU pass, stored) x this "
                                         "run's kernel time; peak = 157.3 TFLOP/s FMA / 2 (assumes packed f32 issue: the "
                                         "4-cycle wave64 issue model tops out at 3.9e13)",
                               "hbm_frac_for_completeness": hbm["frac"]}

    if not args.no_parity:
        out["parity"] = (parity_check_pool(args, local_rank, stream_ptr, K) if args.config == "C5" else
                         parity_check(args, local_rank, stream_ptr, x0, cand, K))

    if not args.no_secondary and world == 1 and args.config == "C2":
        out["secondary"] = secondary(args, local_rank, stream_ptr, x0, B, K, Nh, torch, Engine, N)
        if streamed and args.parts == 1 and Nh * 2 <= 40:
            try:
                out["secondary"]["two_handles"] = two_handles(args, local_rank, x0, cand, B, K, Nh, torch)
            except Exception as e:  # never let a secondary figure take the bench line down
                out["secondary"]["two_handles"] = {"error": str(e)[:300]}

    # the best regime whose candidates CHANGE every tick (the streamed headline re-reads one static tensor)
    cl = closed_loop_value(out.get("secondary") or {})
    if cl:
        out["value_closed_loop"] = cl
    # the two arithmetic widths side by side: the headline is the reference's (float64); the other width of the same tick is
    # this run's secondary.other_width, timed like the headline (in-stream event bracket + dispatch stamps)
    mine = {k: out["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_ms",
                                                "algorithmic_bytes_per_launch")}
    ow = (out.get("secondary") or {}).get("other_width")
    out["value_" + args.dtype], out["roofline_" + args.dtype] = out["value"], mine
    if ow and "env_control_steps_per_s" in ow:
        out["value_" + ow["dtype"]] = ow["env_control_steps_per_s"]
        out["roofline_" + ow["dtype"]] = dict(ow["roofline"], kernel=ow["kernel"], avg_launch_ms=ow["kernel_avg_ms"],
                                              note=f"the same tick with {ow['dtype']} storage and arithmetic, this run "
                                                   "(secondary.other_width)")

    if not args.no_cpu_baseline and world == 1 and args.config == "C2":
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        try:
            out["cpu_baseline"]["reference_algorithm"] = cpu_reference_algorithm(args, min(args.cpu_seconds, 8.0))
        except ImportError as e:  # SciPy missing on the box: the port above is the baseline
            out["cpu_baseline"]["reference_algorithm"] = {"error": str(e)}
    stdout_back()
    print(json.dumps(strict_json(out)))
    sys.stdout.flush()
    if dist is not None:
        stdout_to_stderr()  # (whatever the teardown prints)
        dist.barrier()
        dist.destroy_process_group()
    if "parity" in out and not out["parity"]["ok"]:
        sys.exit("bench.py: the parity check of this run FAILED - the numbers above are not valid")


def closed_loop_value(sec):
    """Top-level ``value_closed_loop``: the best of this run's regimes in which every tick decides over candidates that did
    not exist the tick before - the on-device optimiser (rcg_control_tick_opt), the device-side search
    (rcg_control_tick_search), the produced stream (k_cand_sample rewrites the tensor, then the streamed tick reads it) -
    named, with all of them listed.  The generated level grid is listed too but not eligible: its candidates are the same
    constant sequences every tick."""
    regimes = {}

    def put(name, v, what):
        if isinstance(v, (int, float)) and v > 0:
            regimes[name] = {"value": float(v), "what": what}

    ot = sec.get("optimizer_tick") or {}
    put("optimizer_memory0", (ot.get("memory_0") or {}).get("env_control_steps_per_s"),
        "rcg_control_tick_opt, 5 iterations of k_actor_opt from the warm start, MPC default (no curvature pairs)")
    put("optimizer_memory4", (ot.get("memory_4") or {}).get("env_control_steps_per_s"),
        "the same with 4 curvature pairs (the critic modes' default)")
    ds_ = sec.get("device_search") or {}
    put("device_search_1_round", (ds_.get("rounds_1") or {}).get("env_control_steps_per_s"),
        "rcg_control_tick_search, K candidates drawn and evaluated on the device, 1 round")
    put("device_search_4_rounds", (ds_.get("rounds_4") or {}).get("env_control_steps_per_s"), "the same, 4 refinement rounds")
    put("produced_stream", (sec.get("produced_stream") or {}).get("env_control_steps_per_s"),
        "k_cand_sample rewrites the [B][K][N][du] tensor every tick, then the streamed tick reads it")
    if not regimes:
        return None
    best = max(regimes, key=lambda k: regimes[k]["value"])
    f64r = sec.get("closed_loop_f64") or {}
    f64v = {k: v["env_control_steps_per_s"] for k, v in f64r.items() if isinstance(v, dict) and "env_control_steps_per_s" in v}
    return {"value": regimes[best]["value"], "unit": "env-control-steps/s", "regime": best, "what": regimes[best]["what"],
            "regimes": regimes, "dtype": "f32",
            "value_f64": (max(f64v.values()) if f64v else None), "regime_f64": (max(f64v, key=f64v.get) if f64v else None),
            "regimes_f64": f64v,
            "note": "candidates change every tick in each of these; the headline `value` streams one static tensor"}


def strict_json(x):
    """inf / nan (e.g. the returns of envs whose model blew up and were frozen: `n_failed` > 0) are not JSON: -> None."""
    if isinstance(x, dict):
        return {k: strict_json(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [strict_json(v) for v in x]
    if isinstance(x, float) and (x != x or x in (float("inf"), float("-inf"))):
        return None
    return x


def valu_instr_per_eval(key):
    """VALU instructions per _actor_cost evaluation from the stored SQ_INSTS_VALU pass (profiles/valu_instr.json)."""
    path = os.path.join(ROOT, "profiles", "valu_instr.json")
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path)).get(key)
    except Exception:
        return None


def two_handles(args, device, x0, cand, B, K, Nh, torch):
    """The same tick as TWO handles of B / 2 envs on streams of their own (rcognita_amd.pool.MixedPool(parts=2), `bench.py
    --parts 2`): one part's env step and the ramp-down of its actor kernel overlap the other part's actor kernel."""
    from rcognita_amd.pool import MixedPool

    pool = MixedPool({"3wrobot": B}, device=device, dtype=args.dtype, Nactor=Nh, own_streams=False, parts=2, mode="MPC")
    streams = [torch.cuda.Stream() for _ in pool.segments]
    pool.set_streams([st.cuda_stream for st in streams])
    pool.set_states({"3wrobot": x0})
    torch.cuda.synchronize()
    cd = {"3wrobot": cand}
    for _ in range(300):
        pool.control_tick(K, cd, ordered=True)
    a = [torch.cuda.Event(enable_timing=True) for _ in streams]
    b = [torch.cuda.Event(enable_timing=True) for _ in streams]
    here = [torch.cuda.Event() for _ in streams]  # align the two streams on the device: the timed ticks start together
    for ev, st in zip(here, streams):
        ev.record(st)
    for i, st in enumerate(streams):
        for j, ev in enumerate(here):
            if j != i:
                st.wait_event(ev)
    for ev, st in zip(a, streams):
        ev.record(st)
    n = max(100, args.steps)
    for _ in range(n):
        pool.control_tick(K, cd, ordered=True)
    for ev, st in zip(b, streams):
        ev.record(st)
    torch.cuda.synchronize()
    ms = max(x.elapsed_time(y) for x in a for y in b) / n
    esz = 4 if args.dtype == "f32" else 8
    byt = actor_bytes_per_launch(B, K, Nh, 2, 5, esz, True)
    pool.close()
    return {"env_control_steps_per_s": B / (ms * 1e-3), "ms_per_step": ms, "handles": 2,
            "hbm_frac_at_tick_level": byt / (ms * 1e-3) / HBM_PEAK,
            "note": "aggregate algorithmic bytes of the tick's two actor launches / device time per tick (the launches "
                    "overlap, so a per-launch duration is not meaningful here); `python bench.py --parts 2` makes this the line"}


def secondary(args, device, stream_ptr, x0, B, K, Nh, torch, Engine, N):
    """Other regimes of the same workload on this GPU (rank 0, N = 1 only; not part of `value`)."""
    sec = {}
    du, ds = 2, 5
    # the headline's width and the width of the regimes below: generated grid, device search, optimiser, small batches are
    # float32 regimes (k_ticks_pk / the packed rollouts exist in float32 only; their stored instruction counts are float32's);
    # they are measured in float32 whatever the headline's width, and say so
    head_dtype = args.dtype
    args = argparse.Namespace(**vars(args))
    args.dtype = "f32"
    sec["regimes_dtype"] = "f32 (generated_grid, sim_step_only, device_search, produced_stream, optimizer_tick, small_batch_ticks)"
    # (1) generated level-grid candidates (VALU-bound regime) at the same K
    ecfg, bnds = c2_engine_config(args, device, B)
    eng2 = Engine(ecfg)
    eng2.set_stream(stream_ptr)
    eng2.set_state(x0)
    for _ in range(5):
        eng2.control_tick(None, K=K)
    # wall rate first, un-sampled (a launch that carries start / stop events costs the stream ~7 us: 10 % of this tick),
    # then the kernel's own duration on every launch of a second loop
    # (device time between two in-stream events behind a short spin, as the headline is taken: round 4 read the host clock
    # around 100 ticks from an empty queue, which charged the launch ramp and the final synchronize - 1 to 2 of 5 ms - to the rate)
    n2 = max(400, args.steps // 2)
    st2 = torch.cuda.current_stream()  # the stream the engines were given (main() passes its pointer)
    assert st2.cuda_stream == stream_ptr, (st2.cuda_stream, stream_ptr)
    ea2, eb2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(100):
        eng2.control_tick(None, K=K)
    ea2.record(st2)
    for _ in range(n2):
        eng2.control_tick(None, K=K)
    eb2.record(st2)
    torch.cuda.synchronize()
    d2 = ea2.elapsed_time(eb2) * 1e-3
    eng2.profile((N.KERNEL_ACTOR,), stride=1)
    for _ in range(max(10, args.steps // 4)):
        eng2.control_tick(None, K=K)
    torch.cuda.synchronize()
    ms2, c2 = eng2.profile_read(N.KERNEL_ACTOR)
    eng2.profile(False)
    evals_kernel = B * K / max(ms2 / max(c2, 1) * 1e-3, 1e-12)
    g = {"env_control_steps_per_s": B * n2 / d2, "actor_cost_evals_per_s": B * n2 * K / d2, "bound": "valu",
         "kernel_avg_ms": ms2 / max(c2, 1), "kernel_evals_per_s": evals_kernel}
    ipe = valu_instr_per_eval(f"k_actor_generated_3wrobot_N{Nh}_{args.dtype}")
    if ipe:
        g["roofline_valu"] = {"instr_per_eval": ipe["valu_instr_per_eval"], "lane_instr_per_s": evals_kernel * ipe["valu_instr_per_eval"],
                              "peak": VALU_PEAK, "frac": evals_kernel * ipe["valu_instr_per_eval"] / VALU_PEAK,
                              "source": "profiles/valu_instr.json (rocprofv3 --pmc SQ_INSTS_VALU pass, stored) x this "
                                        "run's kernel time"}
    sec["generated_grid"] = g
    # (2) pure env step: RK4 of closed_loop_rhs only (Simulator.sim_step), 64 B/env algorithmic
    for _ in range(5):
        eng2.sim_step(1)
    eng2.profile((N.KERNEL_SIM,))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n3 = 200
    for _ in range(n3):
        eng2.sim_step(1)
    torch.cuda.synchronize()
    d3 = time.perf_counter() - t1
    ms3, c3 = eng2.profile_read(N.KERNEL_SIM)
    eng2.profile(False)
    sec["sim_step_only"] = {"env_steps_per_s_wall": B * n3 / d3, "kernel_avg_us": ms3 / max(c3, 1) * 1e3,
                            "kernel_GBps": B * ((3 * ds + du) * 4 + 4) / max(ms3 / max(c3, 1) * 1e-3, 1e-12) / 1e9}
    # (2b) a closed loop whose candidates CHANGE every tick (the headline re-reads one static tensor).
    #   device_search: k_actor_search - every lane generates its candidate row (Philox) where it evaluates it, the wave
    #                  refines around the winner; nothing but state, action and cost touches HBM (VALU-bound);
    #   produced_stream: the producer alone (k_cand_sample) writes this tick's [B][K][N][du] tensor, then the streamed tick
    #                  reads it: 2 x 1.34 GB of HBM traffic per tick;
    #   optimizer_tick: the on-device quasi-Newton optimiser as the decision (no candidates at all).
    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    try:
        nn = max(10, args.steps // 10)
        ds_ = {}
        for rounds in (1, 4):
            dt_s = timed(lambda: eng2.control_tick_search(K=K, rounds=rounds, warm_start=True), nn)
            ds_[f"rounds_{rounds}"] = {"env_control_steps_per_s": B / dt_s, "ms_per_step": dt_s * 1e3,
                                       "actor_cost_evals_per_s": B * K * rounds / dt_s}
        ds_["kernel"] = eng2.last_launch(N.KERNEL_ACTOR)["kernel"]
        ds_["bound"] = "valu (Philox + Box-Muller + rollout per candidate; no candidate bytes in HBM)"
        sec["device_search"] = ds_
        if args.regime == "streamed":
            td = torch.float32 if args.dtype == "f32" else torch.float64
            cbuf = torch.empty((B, K, Nh, du), device="cuda", dtype=td)
            torch.cuda.synchronize()

            def produced():
                eng2.candidates_sample(K, round=1, out=cbuf)
                eng2.control_tick(cbuf, K=K)

            dt_p = timed(produced, nn)
            byt = 2 * B * K * Nh * du * (4 if args.dtype == "f32" else 8)
            sec["produced_stream"] = {"env_control_steps_per_s": B / dt_p, "ms_per_step": dt_p * 1e3,
                                      "hbm_GBps_candidates_written_and_read": byt / dt_p / 1e9,
                                      "note": "k_cand_sample writes the tick's candidate tensor, k_actor_dma reads it back"}
            del cbuf
        ot = {}
        for mem in (4, 0):
            eng2.set_optimizer(mem)
            dt_o = timed(lambda: eng2.control_tick_opt(iters=5, warm_start=True), nn)
            ot[f"memory_{mem}"] = {"env_control_steps_per_s": B / dt_o, "ms_per_step": dt_o * 1e3, "iters": 5}
        eng2.set_optimizer(4)
        sec["optimizer_tick"] = ot
        # the same decisions in float64 (the reference's width, the headline's): optimiser tick and device search
        ecfg64, _ = c2_engine_config(args, device, B, dtype="f64")
        e64 = Engine(ecfg64)
        e64.set_stream(stream_ptr)
        e64.set_state(x0)
        f64r = {}
        for mem in (0, 4):
            e64.set_optimizer(mem)
            dt_o = timed(lambda: e64.control_tick_opt(iters=5, warm_start=True), nn)
            f64r[f"optimizer_memory{mem}"] = {"env_control_steps_per_s": B / dt_o, "ms_per_step": dt_o * 1e3, "iters": 5}
        dt_s = timed(lambda: e64.control_tick_search(K=K, rounds=1, warm_start=True), nn)
        f64r["device_search_1_round"] = {"env_control_steps_per_s": B / dt_s, "ms_per_step": dt_s * 1e3}
        e64.close()
        sec["closed_loop_f64"] = f64r
    except Exception as e:  # never let a secondary figure take the bench line down
        sec["device_search"] = {"error": str(e)[:300]}
    eng2.close()
    # (2c) small batches: T ticks per launch (rcg_control_tick_n / rcg_control_ticks) against the loop of single ticks, the
    #   regime where two launches per tick (~8 us) dwarf ~1 us of work: B = 1024, K = 64 of the same workload
    try:
        Bs, Ks, Ts = 1024, 64, 256
        sb = {}
        for tag in ("streamed", "generated"):
            ecs, _ = c2_engine_config(args, device, Bs)
            es = Engine(ecs)
            es.set_stream(stream_ptr)
            es.set_state(x0[:Bs])
            cs = None
            if tag == "streamed":
                blo32 = torch.tensor(bnds[:, 0], device="cuda", dtype=torch.float32 if args.dtype == "f32" else torch.float64)
                bhi32 = torch.tensor(bnds[:, 1], device="cuda", dtype=blo32.dtype)
                cs = (torch.rand((Bs, Ks, Nh, du), device="cuda", dtype=blo32.dtype) * (bhi32 - blo32) + blo32).contiguous()
            per = timed(lambda: [es.control_tick(cs, K=Ks) for _ in range(Ts)], 2) / Ts
            k_per = es.last_launch(N.KERNEL_ACTOR)["kernel"]
            one = timed(lambda: es.control_tick(cs, K=Ks, T=Ts), 4) / Ts
            sb[tag] = {"one_launch_steps_per_s": Bs / one, "per_tick_launches_steps_per_s": Bs / per, "speedup": per / one,
                       "kernel_one_launch": es.last_launch(N.KERNEL_ACTOR)["kernel"], "kernel_per_tick": "k_sim + " + k_per}
            es.close()
        sb["shape"] = f"B={Bs}, K={Ks}, Nactor={Nh}, T={Ts} ticks per call"
        sec["small_batch_ticks"] = sb
    except Exception as e:  # never let a secondary figure take the bench line down
        sec["small_batch_ticks"] = {"error": str(e)[:300]}
    # (3) the OTHER arithmetic width of the streamed tick (headline float64 -> float32 here, and the reverse), timed as the
    # headline is: a spin on the new tensor, then n ticks between two in-stream events, the kernel's own duration from the
    # stamps carried by every 4th dispatch
    if args.regime == "streamed":
        od = "f32" if head_dtype == "f64" else "f64"
        try:
            td = torch.float32 if od == "f32" else torch.float64
            oesz = 4 if od == "f32" else 8
            ecfg_o, _ = c2_engine_config(args, device, B, dtype=od)
            eo = Engine(ecfg_o)
            eo.set_stream(stream_ptr)
            eo.set_state(x0)
            gen = torch.Generator(device="cuda")
            gen.manual_seed(4321)
            blo = torch.tensor(bnds[:, 0], device="cuda", dtype=td)
            bhi = torch.tensor(bnds[:, 1], device="cuda", dtype=td)
            co = (torch.rand((B, K, Nh, du), generator=gen, device="cuda", dtype=td) * (bhi - blo) + blo).contiguous()
            for _ in range(150):  # the tensor is new to the GPU: same clock / TLB warm-up as the main line
                eo.control_tick(co, K=K)
            eo.profile((N.KERNEL_ACTOR,), stride=4)
            n4 = max(40, args.steps // 2)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record(st2)
            for _ in range(n4):
                eo.control_tick(co, K=K)
            eb.record(st2)
            torch.cuda.synchronize()
            d4 = ea.elapsed_time(eb) * 1e-3
            ms4, c4 = eo.profile_read(N.KERNEL_ACTOR)
            ko = eo.last_launch(N.KERNEL_ACTOR)["kernel"]
            eo.close()
            bo = actor_bytes_per_launch(B, K, Nh, du, ds, oesz, True)
            sec["other_width"] = {"env_control_steps_per_s": B * n4 / d4, "ms_per_step": d4 / n4 * 1e3, "dtype": od,
                                  "steps": n4, "clock": "HIP events recorded in-stream around the steps (as the headline)",
                                  "kernel": ko + ("<float>" if od == "f32" else "<double>"),
                                  "kernel_avg_ms": ms4 / max(c4, 1), "launches_timed": int(c4),
                                  "roofline": {"bound": "hbm", "achieved": bo / (ms4 / max(c4, 1) * 1e-3) / 1e9,
                                               "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                               "frac": bo / (ms4 / max(c4, 1) * 1e-3) / HBM_PEAK,
                                               "algorithmic_bytes_per_launch": bo}}
            del co
        except Exception as e:  # never let a secondary figure take the bench line down
            sec["other_width"] = {"error": str(e)[:300]}
    return sec


if __name__ == "__main__":
    main()
