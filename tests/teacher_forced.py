"""Teacher-forced replay of the reference's critic-mode closed loops (fixtures F7c, oracle/gen_critic_fixtures.py), shared
by the CPU twin (tests/test_teacher_forced_oracle.py: the oracle's fit and optimiser) and the GPU test
(tests/test_hip_teacher_forced.py: the mirror classes on librcg).

The loop being matched is rcognita/controllers.py:1458-1477 (push both buffers, refit the critic, run the actor).  At
EVERY control tick of the reference's run the decision maker under test is handed exactly what the reference's saw - the
observation, ``state_sys`` (one simulation step behind, App. A-2), both buffers, ``w_critic_prev``, the action pushed into
the buffer - and must return what the reference's SLSQP returned, in the three senses that can be asked of a different
optimiser:

  critic   P(w_dev) <= P(w_ref) (1 + 1e-9),  P(w) = Jc(w) + mu / 2 |w - w_init|^2,  mu = 1e-8 trace(A A^T) / m, on the
           reference's own TD stack (1248-1271): the build-defined fit is the exact minimiser of P over the box (SLSQP's
           own answer depends on where it stops, and on 4 .. 74 % of the robots' ticks it stops AT w_init), so no feasible
           point - SLSQP's included - may have a lower P; and in plain Jc, Jc(w_dev) <= Jc(w_ref) + FIT_TOL Jc(w_init):
           what the Tikhonov term costs in the directions it damps (sigma^2 / trace ~ 1e-8; measured worst 1.34e-2,
           oracle/experiments/fit_mu_study.py: resolving them as SLSQP does costs 100 x the HIP-vs-oracle agreement);
  actor    J(u_dev; w_ref) <= J(u_ref; w_ref) (1 + ACTOR_TOL) (+ 1e-7)  the weights forced to the reference's (1330-1427);
  action   |u_dev[0, i] - u_ref[0, i]| <= tau_i                      on every tick and component where the reference's own
           cost, with that component pinned tau_i away from SLSQP's optimum and everything else re-optimised by SLSQP,
           rises by more than the excess cost g = J(u_dev) / J* - 1 the decision under test actually left (+ RISE_MARGIN for
           the re-optimisation's own noise) - fixture field tick_first_rise; tau_i = the smallest of 1, 2, 5, 10, 20 % of the
           bound width from which on that holds.  A sequence whose cost is within g of SLSQP's cannot have its first
           action further out than that level set of the reference's own cost profile; directions along which the cost
           is flat (the 3-wheel robot's preset puts no weight on the inputs; on the tank a 20 % move of the first input
           changes the 10-step cost by 0.2 %) are thereby excluded by MEASUREMENT, not by a band.  With g at the granted
           ACTOR_TOL this is the rule "assert where the rise exceeds 0.5 %"; a tighter decision is held tighter.

J and Jc are evaluated by the oracle's operators, which reproduce the reference's on these very ticks to 1e-11
(tests/test_critic_traces.py::test_oracle_operators_on_every_tick_of_the_reference_loop).
"""
import numpy as np

from oracle import rcg_oracle as O

ACTOR_TOL = 0.005  # the optimiser's bar on F8 / F8c: within 0.5 % of SLSQP's cost
ACTOR_ABS = 1e-7   # the accuracy the reference itself asks of SLSQP (tol=1e-7, controllers.py:1396) and the mirror classes'
                   # opt_ftol: two costs closer than that are the same cost (one tick of the NI robot's SQL run has J* = 1.1e-10)
FIT_TOL = 2e-2     # of Jc(w_init): what the fit's Tikhonov term (mu = 1e-8 trace / m) may leave above SLSQP's Jc
RISE_MARGIN = 1e-4  # added to the achieved excess cost: SLSQP's re-optimised profile is itself only good to ~1e-5


def first_action_tau(first_rise, fracs, width, excess):
    """Per input component: the smallest displacement (in input units) from which on the reference's cost rises by more
    than ``excess`` + RISE_MARGIN - inf where it never does (a flat direction: nothing is asserted)."""
    du, nf = first_rise.shape
    tau = np.full(du, np.inf)
    thr = max(float(excess), 0.0) + RISE_MARGIN
    for i in range(du):
        for j in range(nf):
            if np.all(first_rise[i, j:] > thr):
                tau[i] = fracs[j] * width[i]
                break
    return tau


class Tally:
    def __init__(self, what):
        self.what, self.n_ticks, self.n_fits, self.n_sharp, self.n_comp = what, 0, 0, 0, 0
        self.worst_jc, self.worst_j, self.worst_a, self.worst_p = -np.inf, -np.inf, 0.0, -np.inf
        self.failures = []

    def line(self):
        return (f"TEACHER-FORCED {self.what}: {self.n_ticks} ticks, {self.n_fits} fits; worst (Jc_dev - Jc*) / Jc_init "
                f"{self.worst_jc:+.2e}, worst P(w_dev) / P(w_ref) - 1 {self.worst_p:+.2e}; worst J(u_dev; w_ref) / J* - 1 {self.worst_j:+.3%}; first action asserted on "
                f"{self.n_sharp} of {self.n_comp} (tick, component) pairs, worst |du| / tau {self.worst_a:.3f}")


def check_tick(tally, cfg, z, i, w_dev, u_dev, fracs, p_tol=1e-9):
    """``w_dev``: the weights the fit under test returned for tick ``i`` (None if the reference did not refit there);
    ``u_dev [N, du]``: the sequence the actor under test returned with the reference's weights; ``p_tol``: relative slack of the
    fit-objective comparison (1e-9 for float64 arithmetic; a float32 handle stores buffers and weights in float32)."""
    obs, xs = z["tick_obs"][i], z["tick_state_sys"][i]
    tally.n_ticks += 1
    if w_dev is not None:
        tally.n_fits += 1
        jc = float(O.critic_cost(w_dev[None], z["tick_w_prev"][i][None], z["tick_obs_buf"][i][None],
                                 z["tick_act_buf"][i][None], cfg)[0])
        jc_ref, jc0 = float(z["tick_Jc"][i]), float(z["tick_Jc_init"][i])
        tally.worst_jc = max(tally.worst_jc, (jc - jc_ref) / max(jc0, 1e-300))
        if not jc <= jc_ref * (1 + 1e-6) + FIT_TOL * jc0 + 1e-12:
            tally.failures.append(f"tick {i}: Jc {jc:.6g} above the reference's {jc_ref:.6g} (Jc(w_init) {jc0:.4g})")
        # the build-defined objective: the device's weights against SLSQP's
        A, _ = O.critic_td_system(z["tick_w_prev"][i][None], z["tick_obs_buf"][i][None], z["tick_act_buf"][i][None], cfg)
        mu = O.FIT_MU_REL * float(np.sum(A[0] * A[0])) / A.shape[1]
        w0 = np.ones(cfg.dc)
        p_dev = jc + 0.5 * mu * float(np.sum((w_dev - w0) ** 2))
        p_ref = jc_ref + 0.5 * mu * float(np.sum((z["tick_w"][i] - w0) ** 2))
        tally.worst_p = max(tally.worst_p, (p_dev - p_ref) / max(p_ref, 1e-300))
        if not p_dev <= p_ref * (1 + p_tol) + 1e-12:
            tally.failures.append(f"tick {i}: regularised objective {p_dev:.10g} above its value at SLSQP's weights {p_ref:.10g}")
        lo, hi = O.critic_bounds(cfg.critic_struct, cfg.dc)
        if np.any(w_dev < lo - 1e-9) or np.any(w_dev > hi + 1e-9):
            tally.failures.append(f"tick {i}: weights outside [Wmin, Wmax]")
    w_ref = z["tick_w"][i]
    J = float(O.actor_cost(u_dev[None], obs, xs, cfg, w_critic=w_ref)[0])
    J_ref = float(z["tick_J"][i])
    gap = (J - J_ref) / abs(J_ref) if J_ref != 0 and abs(J - J_ref) > ACTOR_ABS else 0.0
    tally.worst_j = max(tally.worst_j, gap)
    if not gap <= ACTOR_TOL:
        tally.failures.append(f"tick {i}: J {J:.8g} vs SLSQP's {J_ref:.8g} ({gap:+.3%})")
    width = cfg.ctrl_bnds[:, 1] - cfg.ctrl_bnds[:, 0]
    tau = first_action_tau(z["tick_first_rise"][i], fracs, width, gap)
    u_ref0 = z["tick_action_sqn"][i][: cfg.du]
    for c in range(cfg.du):
        tally.n_comp += 1
        if not np.isfinite(tau[c]):
            continue
        tally.n_sharp += 1
        d = abs(float(u_dev[0, c]) - float(u_ref0[c])) / tau[c]
        tally.worst_a = max(tally.worst_a, d)
        if d > 1.0:
            tally.failures.append(f"tick {i}: first action component {c}: {u_dev[0, c]:.6g} vs the reference's {u_ref0[c]:.6g}, "
                                  f"further than tau = {tau[c]:.4g} although the reference's cost rises by "
                                  f"{np.round(z['tick_first_rise'][i][c], 4)} along it")
