#!/usr/bin/env python3
"""bench.py - env.control-steps/sec of the rcognita hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): Sys3WRobot, B = 65536 envs per GPU, RK4
dt = 0.01 (one substep per control tick), CtrlOptPred MPC, Nactor = 10, K = 256 candidate action
sequences per env streamed from HBM as a [B][K][N][du] f32 tensor (the `_actor_cost(action_sqn, obs)`
operator shape).  One "step" = one env.control-step (unit U2) for every env of the batch:
rcg_control_tick = k_sim (closed_loop_rhs under RK4) + k_actor (K rollouts + argmin + accum update).
Inputs are synthetic and resident in HBM before the timed region.

One JSON line on stdout (rank 0).  `roofline` prices the dominant kernel k_actor: algorithmic bytes
per launch (DESIGN.md "Bytes") / its mean duration measured with HIP events on the engine's stream
inside the timed region.  `cpu_baseline` is the C oracle (oracle/oracle.c, kind "port") timed on this
box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec; 6.29e12 measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=500)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--batch", type=int, default=65536, help="envs per GPU (weak scaling)")
    p.add_argument("--candidates", type=int, default=256, help="K candidate sequences per env")
    p.add_argument("--nactor", type=int, default=10)
    p.add_argument("--regime", choices=["streamed", "generated"], default="streamed")
    p.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                   help="nccl = RCCL over xGMI (the real thing); gloo only to exercise the N>1 code path on one GPU")
    p.add_argument("--single-device", action="store_true",
                   help="debug: every rank uses cuda:0 (with --dist-backend gloo), to test the N>1 path on a 1-GPU box")
    p.add_argument("--profile-stride", type=int, default=8,
                   help="bracket every n-th kernel launch of the timed region with HIP events (0 = none)")
    return p.parse_args()


def make_engine(args, device, batch=None):
    from rcognita_amd import Engine, EngineConfig
    from rcognita_amd import _native as N

    bnds = np.array([[-300.0, 300.0], [-100.0, 100.0]])  # presets/main_3wrobot.py:207-211
    R1 = np.diag([1.0, 10.0, 1.0, 0, 0, 0, 0])  # presets/main_3wrobot.py R1_diag default
    cfg = EngineConfig(sys_id=N.SYS_3WROBOT, batch=batch or args.batch, dtype=args.dtype, device=device,
                       Nactor=args.nactor, mode="MPC", pars=[10.0, 1.0], ctrl_bnds=bnds, R1=R1, gamma=1.0,
                       dt_sim=0.01, sampling_time=0.01, pred_step_size=0.02, substeps_per_tick=1)
    return Engine(cfg), bnds, R1


def synth_state(rank, B):
    """seed 1234 + rank; x,y ~ U(-10,10), alpha ~ U(-pi,pi), v, omega ~ U(-1,1)."""
    rng = np.random.default_rng(1234 + rank)
    return np.stack([rng.uniform(-10, 10, B), rng.uniform(-10, 10, B), rng.uniform(-np.pi, np.pi, B),
                     rng.uniform(-1, 1, B), rng.uniform(-1, 1, B)], axis=-1)


def actor_bytes_per_launch(B, K, N, du, ds, esz, streamed, fused_sim):
    """Algorithmic HBM bytes of one actor launch in tick mode (DESIGN.md 'Bytes')."""
    per_env = ds * esz  # state read (obs == state_sys)
    per_env += du * esz + esz + 4  # action, best_J, best_idx writes
    per_env += 2 * esz + 2 * 4  # accum and step_idx read-modify-write
    if streamed:
        per_env += K * N * du * esz  # the candidate rows
    if fused_sim:  # env step inside the launch: held action + status read, state + state_prev written
        per_env += du * esz + 4 + 2 * ds * esz
    return B * per_env


def cpu_baseline(args, seconds):
    """C oracle (port of the same algorithm, f64) on the host cores, bounded sample of the workload."""
    from oracle import c_oracle as CO
    from oracle import rcg_oracle as O

    threads = CO.max_threads()
    bnds = np.array([[-300.0, 300.0], [-100.0, 100.0]])
    cfg = O.OracleCfg(sys_id=O.SYS_3WROBOT, n_actor=args.nactor, pars=[10.0, 1.0], ctrl_bnds=bnds,
                      R1=np.diag([1.0, 10.0, 1.0, 0, 0, 0, 0]), gamma=1.0, dt_sim=0.01, sampling_time=0.01,
                      pred_step_size=0.02)
    K = args.candidates
    Bc = 256 * threads
    rng = np.random.default_rng(99)
    cb = CO.CBatch(cfg, synth_state(0, Bc))
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
    return {"value": n / dt, "unit": "env-control-steps/s", "cores": threads, "kind": "port",
            "sample": f"{Bc} envs x {ticks + 1} ticks, K={K}, Nactor={args.nactor}, C oracle f64 + OpenMP, {dt:.1f} s"}


def cpu_reference_algorithm(args, seconds):
    """The reference ALGORITHM on one host core: one env, SciPy RK45 + SLSQP over the oracle's operators
    (oracle/ref_loop.py, pinned on traces captured from the reference).  Bounded by wall time."""
    from oracle import rcg_oracle as O
    from oracle.ref_loop import RefLoop

    bnds = np.array([[-300.0, 300.0], [-100.0, 100.0]])
    cfg = O.OracleCfg(sys_id=O.SYS_3WROBOT, n_actor=args.nactor, pars=[10.0, 1.0], ctrl_bnds=bnds,
                      R1=np.diag([1.0, 10.0, 1.0, 0, 0, 0, 0]), gamma=1.0, dt_sim=0.01, sampling_time=0.01,
                      pred_step_size=0.02)
    loop = RefLoop(cfg, np.array([5.0, 5.0, -3 * np.pi / 4, 0.0, 0.0]), t1=1e9)
    t0 = time.perf_counter()
    ticks, last = 0, None
    while time.perf_counter() - t0 < seconds:
        row = loop.step()
        act = tuple(row[1 + cfg.ds:1 + cfg.ds + cfg.du])
        if last is not None and act != last:
            ticks += 1
        last = act
    dt = time.perf_counter() - t0
    return {"value": ticks / dt, "unit": "env-control-steps/s", "cores": 1, "kind": "reference algorithm "
            "(SciPy RK45 + SLSQP over the oracle's operators, preset main_3wrobot)",
            "sample": f"1 env, {ticks} control ticks, {loop.nfev_actor} _actor_cost evaluations, {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch  # device memory for the synthetic candidates, stream, torch.distributed (plumbing)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    coll_dev = torch.device("cuda", local_rank)  # where collective payloads live
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=coll_dev)
        else:
            coll_dev = torch.device("cpu")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from rcognita_amd import _native as N
    from rcognita_amd.parallel import gather_summaries

    B, K, Nh, du, ds = args.batch, args.candidates, args.nactor, 2, 5
    eng, bnds, _ = make_engine(args, local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_state(synth_state(rank, B))
    tdtype = torch.float32 if args.dtype == "f32" else torch.float64
    esz = 4 if args.dtype == "f32" else 8
    cand = None
    if args.regime == "streamed":
        g = torch.Generator(device="cuda")
        g.manual_seed(1234 + rank)
        lo = torch.tensor(bnds[:, 0], device="cuda", dtype=tdtype)
        hi = torch.tensor(bnds[:, 1], device="cuda", dtype=tdtype)
        cand = torch.rand((B, K, Nh, du), generator=g, device="cuda", dtype=tdtype) * (hi - lo) + lo
        cand = cand.contiguous()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.control_tick(cand, K=K)
    returns_dev = torch.empty(B, device="cuda", dtype=tdtype)
    gathered = [torch.empty(B, device=coll_dev, dtype=tdtype) for _ in range(world)] if dist is not None else None

    # HIP events around the kernels of the tick on the engine's own stream, inside the timed region; sampled
    # (every n-th launch) because each event is a marker packet on the stream.
    if args.profile_stride > 0:
        if args.steps // args.profile_stride < 8:  # short runs: time (almost) every launch rather than none
            args.profile_stride = max(1, args.steps // 8)
        eng.profile((N.KERNEL_ACTOR, N.KERNEL_SIM), stride=args.profile_stride)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.control_tick(cand, K=K)
    if dist is not None:
        # episode-end exchange: ONE all_gather of the per-env running returns over RCCL
        N.check(N.lib().rcg_get_field(eng._h, N.FIELD_ACCUM, returns_dev.data_ptr(), N.DEVICE), eng._h)
        dist.all_gather(gathered, returns_dev if coll_dev.type == "cuda" else returns_dev.cpu())
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    actor_ms, actor_n = eng.profile_read(N.KERNEL_ACTOR)
    sim_ms, sim_n = eng.profile_read(N.KERNEL_SIM)
    eng.profile(False)

    summ, _ = eng.episode_stats(from_accum=True)
    total = gather_summaries(summ, dist, device=coll_dev)  # per-shard summaries -> whole-job summary
    steps_idx = eng.get_field(N.FIELD_STEP_IDX)
    assert int(steps_idx.min()) == int(steps_idx.max()) == args.warmup + args.steps, "step counter mismatch"

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    units = world * B * args.steps
    value = units / dt
    streamed = args.regime == "streamed"
    fused_sim = False  # the env step is its own launch (k_sim); kept as a parameter of the byte model
    bytes_launch = actor_bytes_per_launch(B, K, Nh, du, ds, esz, streamed, fused_sim)
    actor_avg_s = (actor_ms / max(actor_n, 1)) * 1e-3
    achieved = bytes_launch / actor_avg_s if actor_avg_s > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(f"k_actor_{args.regime}_B{B}_K{K}_N{Nh}_{args.dtype}", {}).get(
                "hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "env-control-steps/sec (whole node), 3wrobot Nactor=10",
        "value": value,
        "unit": "env-control-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"Sys3WRobot B={B}/GPU RK4 dt=0.01 S=1, CtrlOptPred MPC Nactor={Nh}, "
                               f"K={K} {args.regime} candidates (BASELINE configs[1])",
                   "envs_per_gpu": B, "candidates": K, "nactor": Nh, "regime": args.regime,
                   "parallelism": f"env-shard x{world}", "actor_cost_evals_per_s": value * K},
        "roofline": {"bound": "hbm", "kernel": "k_actor_dma" if (streamed and args.dtype == "f32" and K % 64 == 0) else "k_actor",
                     "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": traffic,
                     "algorithmic_bytes_per_launch": bytes_launch, "avg_launch_ms": actor_avg_s * 1e3,
                     "launches_timed": actor_n, "event_stride": args.profile_stride,
                     "sim_kernel_avg_ms": (sim_ms / sim_n) if sim_n else None,
                     "note": ("streamed regime: HBM-bound" if streamed else
                              "generated regime is VALU-bound; the HBM fraction is reported for completeness only")},
        "returns_summary": total,
    }

    if not args.no_secondary and world == 1:
        sec = {}
        # (1) generated level-grid candidates (VALU-bound regime) at the same K
        eng2, _, _ = make_engine(args, local_rank)
        eng2.set_stream(torch.cuda.current_stream().cuda_stream)
        eng2.set_state(synth_state(rank, B))
        for _ in range(5):
            eng2.control_tick(None, K=K)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n2 = max(10, args.steps // 4)
        for _ in range(n2):
            eng2.control_tick(None, K=K)
        torch.cuda.synchronize()
        d2 = time.perf_counter() - t1
        sec["generated_grid"] = {"env_control_steps_per_s": B * n2 / d2, "actor_cost_evals_per_s": B * n2 * K / d2,
                                 "bound": "valu"}
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
                                "kernel_GBps": B * ((3 * ds + du) * esz + 4) / max(ms3 / max(c3, 1) * 1e-3, 1e-12) / 1e9}
        eng2.close()
        out["secondary"] = sec

    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        try:
            out["cpu_baseline"]["reference_algorithm"] = cpu_reference_algorithm(args, min(args.cpu_seconds, 8.0))
        except ImportError as e:  # SciPy missing on the box: the port above is the baseline
            out["cpu_baseline"]["reference_algorithm"] = {"error": str(e)}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
