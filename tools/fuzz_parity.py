#!/usr/bin/env python3
"""Randomised differential run (GPU box): random configurations - system, element type, mode, critic structure, stage-cost structure,
target, discount, horizon, K, batch, TD rows - through the streamed operator / argmin and through closed-loop ticks (env step, buffer
push, critic fit, decision), every number against the oracle.  Complements the parametrised tests: the shapes here are not
hand-picked.   python tools/fuzz_parity.py [cases] [seed]     TEST INFRASTRUCTURE: the oracle is the checker."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import rcg_oracle as O  # noqa: E402
from tests.helpers import PRESETS, both, rand_actions, rand_states, rel_err_norm  # noqa: E402



def run(n_cases, seed, verbose=False):
    """-> (failures [str], worst relative error per element type, kernels that served the streamed decisions)"""
    rng = np.random.default_rng(seed)
    TOL = {"f64": 1e-10, "f32": 2e-5}
    worst = {"f64": 0.0, "f32": 0.0}
    kernels = {}
    fails = []
    t0 = time.time()
    for case in range(n_cases):
        name = str(rng.choice(["3wrobot", "3wrobotNI", "2tank"]))
        dtype = str(rng.choice(["f64", "f32"]))
        mode = str(rng.choice(["MPC", "MPC", "RQL", "SQL"]))
        cs = str(rng.choice(["quad-lin", "quadratic", "quad-nomix", "quad-mix"]))
        N = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 16, 20, 24]))
        K = int(rng.choice([1, 2, 5, 8, 16, 31, 32, 33, 40, 64, 65, 100, 128, 200, 256, 300]))
        B = int(rng.choice([1, 2, 3, 7, 16, 17, 33, 64, 70]))
        huge = bool(rng.uniform() < 0.12)  # a batch of several waves / blocks: the operator, the argmin and the grid only (the oracle's loops are slow)
        if huge:
            B = int(rng.choice([257, 1000, 4097, 8200]))
            K = min(K, 100)
        p = PRESETS[name]
        n = len(p["R1"])
        kw = dict(n_actor=N, gamma=float(rng.choice([1.0, 0.97, 0.9])), mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs])
        stage = str(rng.choice(["diag", "diag", "full", "biquad", "biquad-full"]))
        if stage in ("full", "biquad-full"):
            M = rng.normal(size=(n, n)) * 0.3
            kw["R1"] = M @ M.T + np.diag(p["R1"])
        if stage.startswith("biquad"):
            kw["stage_obj_struct"] = O.STAGE_IDS["biquadratic"]
            kw["R2"] = np.diag(rng.uniform(0.0, 0.5, n)) if stage == "biquad" else (lambda Q: Q @ Q.T)(rng.normal(size=(n, n)) * 0.2)
        if rng.uniform() < 0.3:
            kw["target"] = list(rng.uniform(-1, 1, len(p["x0"])))
        if mode != "MPC":
            bs = int(rng.choice([4, 6, 10, 14]))
            kw.update(buffer_size=bs, n_critic=int(rng.integers(2, min(bs, 13))))
        what = f"case {case}: {name} {dtype} {mode} {cs} N={N} K={K} B={B} stage={stage} target={'target' in kw} " + (f"bs={kw.get('buffer_size')} Nc={kw.get('n_critic')}" if mode != "MPC" else "")
        try:
            eng, cfg = both(name, B, dtype, **kw)
            x = rand_states(rng, name, B)
            cand = rand_actions(rng, name, (B, K, N), overshoot=1.2)
            eng.set_state(x)
            from rcognita_amd import _native as Nn

            w = None
            if mode != "MPC":
                lo, hi = O.critic_bounds(cfg.critic_struct, cfg.dc)
                w = rng.uniform(np.maximum(lo, -2.0), np.minimum(hi, 2.0), (B, cfg.dc))
                eng.set_field(Nn.FIELD_W_CRITIC, w)
                eng.set_field(Nn.FIELD_W_PREV, w)
            # 1. the streamed operator and the argmin
            J = eng.actor_cost(cand)
            J_or = O.actor_cost(cand, x[:, None, :], x[:, None, :], cfg, w_critic=None if w is None else w[:, None, :])
            # float64 element by element; float32 against the largest cost of the batch (random signed critic weights make a J a
            # difference of large terms: its float32 error scales with the terms, not with the difference)
            e1 = rel_err_norm(J, J_or) if dtype == "f64" else float(np.max(np.abs(J - J_or)) / max(float(np.max(np.abs(J_or))), 1.0))
            act, bj, bi = eng.actor_argmin(cand)
            ok_idx = np.array_equal(bi, np.argmin(J, axis=1).astype(np.int32))
            ll = eng.last_launch()
            kernels[ll["kernel"]] = kernels.get(ll["kernel"], 0) + 1
            # 2. closed-loop ticks against the oracle's batch (f64: same decisions; f32: costs agree)
            env = O.new_batch(cfg, x)
            if w is not None:
                env.w_critic, env.w_prev = w.copy(), w.copy()
            e2 = 0.0
            T = 0 if huge else (4 if mode == "MPC" else int(kw["buffer_size"]) + 2)
            same_path = True
            for t in range(T):
                eng.control_tick(cand, K=K)
                O.control_tick(cfg, env, cand)
                st = eng.get_state().astype(np.float64)
                if dtype == "f64" and not np.array_equal(eng.get_field(Nn.FIELD_BEST_IDX), env.best_idx):
                    same_path = False  # (a tie within rounding: the trajectories part - compare up to here only)
                    break
                e2 = max(e2, rel_err_norm(st, env.state, floor=1.0))
                if dtype == "f32":
                    break  # float32 decisions may differ by a rounding: one tick is the statement
            # (the ticks above refitted the critic and filled the buffers: the remaining checks start from fresh handles)
            eng.close()

            def fresh():
                e_, _ = both(name, B, dtype, **kw)
                e_.set_state(x)
                if w is not None:
                    e_.set_field(Nn.FIELD_W_CRITIC, w)
                    e_.set_field(Nn.FIELD_W_PREV, w)
                return e_

            eng = fresh()
            # 3. the generated level grid (cand == NULL) against the oracle's grid
            e3 = 0.0
            Kg = int(rng.choice([1, 4, 9, 16, 64, 256])) if cfg.du == 2 else int(rng.choice([1, 3, 8, 64, 100, 256]))
            eng.set_state(x)
            actg, bjg, big = eng.actor_argmin(None, K=Kg)
            grid = O.grid_candidates(cfg, Kg)
            Jg = O.actor_cost(grid[None], x[:, None, :], x[:, None, :], cfg, w_critic=None if w is None else w[:, None, :])
            bjg_or, big_or = O.argmin_first(Jg)
            e3 = rel_err_norm(bjg, bjg_or) if dtype == "f64" else float(np.max(np.abs(bjg - bjg_or)) / max(float(np.max(np.abs(Jg))), 1.0))
            if dtype == "f64" and not np.array_equal(big, big_or):
                # a mirrored pair of grid levels can tie exactly (a cost that only sees u^2: RQL at Nactor = 1): whichever of the
                # two the rounding of the level favours is a correct argmin - anything else is a failure
                sel = Jg[np.arange(B), big]
                if not np.all(sel <= bjg_or + 1e-12 * np.maximum(np.abs(bjg_or), 1.0)):
                    fails.append(f"{what}: generated grid K={Kg}: best_idx differs from the oracle's and is no tie")
            # 4. the on-device optimiser against its oracle twin (float64: the same walk)
            e4 = 0.0
            if dtype == "f64" and B <= 17 and N <= 10 and not huge:
                its = int(rng.integers(1, 5))
                _, U, Jo, _ = eng.actor_optimize(iters=its)
                U_or, J_or2, _ = O.actor_optimize(cfg, x, x, O.action_sqn_init(cfg, None), iters=its, w_critic=w)
                e4 = rel_err_norm(Jo, J_or2)
                if e4 > 1e-8:  # (a discrete line search: a rounding-level tie between two step lengths parts the walks - the cost must still be the twin's)
                    e4 = 0.0 if np.all(Jo <= J_or2 * (1 + 1e-6) + 1e-9) or np.all(np.abs(Jo - J_or2) <= 1e-3 * np.maximum(np.abs(J_or2), 1.0)) else e4
            # 5. T ticks in one native call = T single ticks, bit for bit
            eng.close()
            eng, eng2 = fresh(), fresh()
            Tn = int(rng.integers(2, 5))
            eng.control_tick(cand, K=K, T=Tn)
            for _ in range(Tn):
                eng2.control_tick(cand, K=K)
            if not np.array_equal(eng.get_state(), eng2.get_state()) or not np.array_equal(eng.get_field(Nn.FIELD_ACCUM), eng2.get_field(Nn.FIELD_ACCUM)):
                fails.append(f"{what}: {Tn} ticks in one call differ from {Tn} single ticks")
            eng2.close()
            err = max(e1, e2, e3, e4)
            worst[dtype] = max(worst[dtype], err)
            if not (err <= TOL[dtype]) or not ok_idx:
                fails.append(f"{what}: J {e1:.2e} loop {e2:.2e} grid {e3:.2e} optimiser {e4:.2e} argmin-consistent {ok_idx} ({ll})")
            eng.close()
        except Exception as ex:  # noqa: BLE001
            fails.append(f"{what}: {type(ex).__name__}: {str(ex)[:200]}")
        if verbose and case % 20 == 19:
            print(f"{case + 1} cases, {len(fails)} failures, worst f64 {worst['f64']:.2e} f32 {worst['f32']:.2e}, {time.time() - t0:.0f} s", flush=True)
    return fails, worst, kernels


def run_bits(n_cases, seed, verbose=False):
    """Claims of bit-identity between code paths, no oracle involved, on batches large enough to meet one-in-a-thousand events:
    T ticks in one native call (k_ticks / k_ticks_pk / k_ticks_mem, or the library's own loop) against T single ticks - streamed and
    generated candidates, every mode - on STATE, ACTION, ACCUM, BEST_J, BEST_IDX and the critic's weights and buffers."""
    from rcognita_amd import _native as Nn

    rng = np.random.default_rng(seed)
    fails, kernels = [], {}
    for case in range(n_cases):
        name = str(rng.choice(["3wrobot", "3wrobotNI", "2tank"]))
        dtype = str(rng.choice(["f64", "f32"]))
        mode = str(rng.choice(["MPC", "MPC", "RQL", "SQL"]))
        cs = str(rng.choice(["quad-lin", "quadratic", "quad-nomix", "quad-mix"]))
        N = int(rng.choice([1, 3, 5, 7, 10, 16]))
        B = int(rng.choice([1024, 3000, 4096]))
        generated = bool(rng.uniform() < 0.4)
        du = 1 if name == "2tank" else 2
        K = int(rng.choice([16, 64, 256] if du == 2 else [8, 64, 100, 256])) if generated else int(rng.choice([4, 8, 16, 32, 40, 64, 100]))
        T = int(rng.integers(2, 6))
        kw = dict(n_actor=N, gamma=float(rng.choice([1.0, 0.95])), mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs])
        if mode != "MPC":
            kw.update(buffer_size=int(rng.choice([4, 6, 10])), n_critic=int(rng.choice([2, 3, 4])))
        what = f"bits case {case}: {name} {dtype} {mode} {cs} N={N} K={K} B={B} T={T} {'generated' if generated else 'streamed'}"
        try:
            x = rand_states(rng, name, B)
            cand = None if generated else rand_actions(rng, name, (B, K, N), overshoot=1.1)
            outs = []
            variants = [False, True] + (["halves"] if (mode != "MPC" and not generated and B >= 2048) else [])
            for single in variants:  # "halves": single ticks, each as two halves on two internal streams (rcg_set_tick_parts 2)
                e, cfg = both(name, B, dtype, **kw)
                e.set_state(x)
                if single == "halves":
                    e.set_tick_parts(2)
                dcand = None if cand is None else e.to_device(cand.astype(e.real))
                if single:
                    for _ in range(T):
                        e.control_tick(dcand, K=K)
                else:
                    e.control_tick(dcand, K=K, T=T)
                ll = e.last_launch()
                fields = [Nn.FIELD_STATE, Nn.FIELD_ACTION, Nn.FIELD_ACCUM, Nn.FIELD_BEST_J, Nn.FIELD_BEST_IDX, Nn.FIELD_STEP_IDX]
                if mode != "MPC":
                    fields += [Nn.FIELD_W_CRITIC, Nn.FIELD_W_PREV, Nn.FIELD_OBS_BUF, Nn.FIELD_ACT_BUF]
                outs.append(([e.get_field(f).copy() for f in fields], ll))
                e.close()
            kernels[outs[0][1]["kernel"]] = kernels.get(outs[0][1]["kernel"], 0) + 1
            for other in range(1, len(outs)):
                for fi, (a, b) in enumerate(zip(outs[0][0], outs[other][0])):
                    if fi == 3 and name == "2tank" and np.array_equal(outs[0][0][4], outs[other][0][4]):
                        # rcg.h, rcg_control_ticks: the tank's rollout leaves its fused multiply-adds to the compiler, and the persistent
                        # and the per-tick kernels inline it differently - BEST_J to a rounding of the rollout's terms, same decisions
                        eps = float(np.finfo(a.dtype).eps)  # (a rounding of the rollout's terms: a few ulp of the largest cost around)
                        if np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= 8.0 * eps * max(float(np.nanmax(np.abs(a))), 1.0)):
                            continue
                    if not np.array_equal(a, b, equal_nan=True):
                        nd = int(np.sum(a != b))
                        fails.append(f"{what}: field #{fi} differs in {nd} of {a.size} entries (max {float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))):.2e}); one call: {outs[0][1]}, {'single' if other == 1 else 'single ticks in two halves'}: {outs[other][1]}")
                        break
        except Exception as ex:  # noqa: BLE001
            fails.append(f"{what}: {type(ex).__name__}: {str(ex)[:200]}")
        if verbose and case % 10 == 9:
            print(f"bits: {case + 1} cases, {len(fails)} failures", flush=True)
    return fails, kernels


def run_loop(n_cases, seed, verbose=False):
    """The drop-in classes' loop (tests/test_hip_loop_step.py::_run: the reference's loop body on System / Simulator / CtrlOptPred) for
    random systems, modes, critic structures, horizons, batches and element types: the rows with the next step started ahead, with one
    native call per iteration, and with the separate calls must be the same bits."""
    from tests.test_hip_loop_step import _run

    rng = np.random.default_rng(seed)
    fails = []
    for case in range(n_cases):
        name = str(rng.choice(["3wrobot", "3wrobotNI", "2tank"]))
        mode = str(rng.choice(["MPC", "RQL", "SQL"]))
        cs = str(rng.choice(["quad-lin", "quadratic", "quad-nomix", "quad-mix"]))
        dtype = str(rng.choice(["f64", "f32"]))
        Nactor = int(rng.choice([1, 2, 3, 5, 8]))
        B = None if rng.uniform() < 0.5 else int(rng.choice([2, 3, 5, 9]))
        T = int(rng.integers(6, 26))
        what = f"loop case {case}: {name} {mode} {cs} {dtype} Nactor={Nactor} B={B} T={T}"
        try:
            ahead, c2 = _run(name, mode, cs, True, T, B=B, dtype=dtype, Nactor=Nactor, speculate=True)
            fused, c1 = _run(name, mode, cs, True, T, B=B, dtype=dtype, Nactor=Nactor, speculate=False)
            plain, c0 = _run(name, mode, cs, False, T, B=B, dtype=dtype, Nactor=Nactor)
            if not (np.array_equal(ahead, fused, equal_nan=True) and np.array_equal(ahead, plain, equal_nan=True)):
                bad = np.argwhere(~((ahead == plain) | (np.isnan(ahead) & np.isnan(plain))))
                fails.append(f"{what}: rows differ (first at row/col {bad[0].tolist() if len(bad) else '?'}); hits {c2.spec_hits} drops {c2.spec_drops} fused {c1.fused_steps}")
            elif c2.spec_hits < T - 4:
                fails.append(f"{what}: only {c2.spec_hits} of {T} iterations were started ahead ({c2.spec_drops} dropped)")
        except Exception as ex:  # noqa: BLE001
            fails.append(f"{what}: {type(ex).__name__}: {str(ex)[:200]}")
        if verbose and case % 10 == 9:
            print(f"loop: {case + 1} cases, {len(fails)} failures", flush=True)
    return fails


def run_loop_step(n_cases, seed, verbose=False):
    """rcg_loop_step against the separate calls (set ACTION, rcg_sim_step_h, rcg_critic_update, rcg_actor_optimize, rcg_stage_obj) on a
    twin handle, batches as large as the entry point takes: state, action, stage cost, best_J, weights - the same bits (the claim of
    rcg.h; tests/test_hip_loop_step.py holds it on four shapes of up to 7 envs)."""
    from rcognita_amd import _native as Nn

    rng = np.random.default_rng(seed)
    fails = []
    for case in range(n_cases):
        name = str(rng.choice(["3wrobot", "3wrobotNI", "2tank"]))
        mode = str(rng.choice(["MPC", "MPC", "RQL", "SQL"]))
        cs = str(rng.choice(["quad-lin", "quadratic", "quad-nomix", "quad-mix"]))
        dtype = str(rng.choice(["f64", "f32"]))
        Nactor = int(rng.choice([1, 3, 5, 8]))
        kw = dict(n_actor=Nactor, mode=O.MODE_IDS[mode], critic_struct=O.CRITIC_IDS[cs], n_critic=int(rng.choice([2, 3, 4])), buffer_size=6)
        what = f"loop_step case {case}: {name} {mode} {cs} {dtype} Nactor={Nactor}"
        try:
            probe, cfg = both(name, 1, dtype, **kw)
            row = cfg.ds + cfg.du + 2 + (cfg.dc if mode != "MPC" else 0)
            probe.close()
            B = int(min(200, (16384 // 8 - 1) // (row + cfg.du + 1)))  # what fits the 16-KB pinned buffer
            a, _ = both(name, B, dtype, **kw)
            b, _ = both(name, B, dtype, **kw)
            x0 = rand_states(rng, name, B)
            a.set_state(x0)
            b.set_state(x0)
            act = rand_actions(rng, name, (B,), overshoot=0.9)
            h = cfg.dt_sim / 2
            for it in range(int(rng.integers(6, 14))):
                decide = it % 2 == 1
                push = mode != "MPC" and decide
                fit = push and it % 4 == 1
                st, ac, stage, bj, w = a.loop_step(act, h, 1, decide=decide, push=push, fit=fit, iters=4)
                b.set_field(Nn.FIELD_ACTION, act)
                b.sim_step(1, step=h)
                if push:
                    b.critic_update(do_fit=fit)
                xs = b.get_field(Nn.FIELD_STATE_PREV)
                st_b = b.get_state()
                if decide:
                    a_b, _, bj_b, _ = b.actor_optimize(iters=4, obs=st_b, state_sys=xs)
                    b.set_field(Nn.FIELD_ACTION, a_b)
                else:
                    a_b, bj_b = act.astype(b.real), None
                stage_b = b.stage_obj(st_b, a_b)
                bad = [n for n, u, v in (("state", st, st_b), ("action", ac, a_b), ("stage", stage, stage_b)) if not np.array_equal(u, np.asarray(v, dtype=np.float64))]
                if decide and not np.array_equal(bj, bj_b.astype(np.float64)):
                    bad.append("best_J")
                if mode != "MPC" and not np.array_equal(w, b.get_field(Nn.FIELD_W_CRITIC).astype(np.float64)):
                    bad.append("w_critic")
                if bad:
                    fails.append(f"{what} B={B} iteration {it}: {bad} differ")
                    break
                act = np.asarray(ac, dtype=np.float64).copy()
            a.close()
            b.close()
        except Exception as ex:  # noqa: BLE001
            fails.append(f"{what}: {type(ex).__name__}: {str(ex)[:200]}")
        if verbose and case % 10 == 9:
            print(f"loop_step: {case + 1} cases, {len(fails)} failures", flush=True)
    return fails


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2026
    fails, worst, kernels = run(n_cases, seed, verbose=True)
    print(f"fuzz: {n_cases} cases (seed {seed}), kernels {kernels}, worst relative error f64 {worst['f64']:.2e} / f32 {worst['f32']:.2e}, {len(fails)} failures")
    for f in fails[:30]:
        print("  FAIL", f)
    nb = max(n_cases // 5, 10)
    bfails, bk = run_bits(nb, seed, verbose=True)
    print(f"bit-identity of T ticks per call vs single ticks: {nb} cases on 1 024 - 4 096 envs, one-call kernels {bk}, {len(bfails)} failures")
    for f in bfails[:30]:
        print("  FAIL", f)
    nl = max(n_cases // 10, 10)
    lfails = run_loop(nl, seed, verbose=True)
    print(f"drop-in loop, next step started ahead / one call per iteration / separate calls: {nl} random cases, {len(lfails)} failures")
    for f in lfails[:30]:
        print("  FAIL", f)
    sfails = run_loop_step(nl, seed, verbose=True)
    print(f"rcg_loop_step against the separate calls, up to 200 envs: {nl} random cases, {len(sfails)} failures")
    for f in sfails[:30]:
        print("  FAIL", f)
    sys.exit(1 if fails or bfails or lfails or sfails else 0)
