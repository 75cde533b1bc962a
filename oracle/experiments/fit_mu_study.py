#!/usr/bin/env python3
"""Round-5 study (TEST INFRASTRUCTURE, CPU): what would a smaller Tikhonov weight, or residual-correction steps after the
active-set walk, buy the build-defined critic fit on the TD stacks of the reference's own closed loops (F7c)?

    python oracle/experiments/fit_mu_study.py      # needs a critic_fit_single with FIT_REFINE (kept in git history: the
                                                   # prototype was `w_F += A_F^T (A_F A_F^T + mu I)^-1 (b - A w)` twice)

Measured (354 stacks): mu = 1e-8 / 1e-9 / 1e-10 (relative to trace(A A^T) / m) leave at worst 1.3e-2 / 5.2e-4 / 1.4e-7
Jc(w_init) of residual above SLSQP's Jc - and a relative perturbation of 1e-14 of (A, b), the size of an fma-contraction
difference between the kernel and numpy, moves the fitted weights by up to 2.4e-6 / 2.4e-5 / 2.4e-4 (p99: 8e-7 / 8e-6 /
8e-5): the directions SLSQP resolves and the regularised fit does not have sigma^2 / trace ~ 1e-8, and resolving them costs
exactly that conditioning.  mu = 1e-10 was built and run on the GPU: 19 of the 1017 parity tests fail (HIP vs oracle weights
1.2e-6 .. 1.1e-5 against a 1e-6 tolerance, best_J 1e-7 .. 7e-6, and two argmin flips without a near-tie on rank-deficient
stacks).  Residual correction does not reach those directions either (factor mu / (sigma^2 + mu) = 0.2 per step).  The fit
stays at mu = 1e-8; tests/teacher_forced.py asserts what that fit promises - its own objective at the device's weights is
not above the objective at SLSQP's weights - and reports the plain Jc gap.
"""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import rcg_oracle as O
from tests.conftest import load_golden
from tests.test_critic_traces import trace_cfg, CASES, MODES
stacks=[]
for name,cs in CASES:
    for mode in MODES:
        meta,z=load_golden(f"F7c_trace_{name}_{mode}_{cs}"); cfg=trace_cfg(meta)
        A,b=O.critic_td_system(z["tick_w_prev"],z["tick_obs_buf"],z["tick_act_buf"],cfg)
        lo,hi=O.critic_bounds(cfg.critic_struct,cfg.dc)
        for i in range(len(A)):
            stacks.append((A[i],b[i],lo,hi,z["tick_Jc"][i],z["tick_Jc_init"][i]))
rng=np.random.default_rng(0)
pert=[(A*(1+1e-14*rng.standard_normal(A.shape)), b*(1+1e-14*rng.standard_normal(b.shape))) for A,b,*_ in stacks]
def jc(A,b,w): r=A@w-b; return 0.5*r@r
for mu in (1e-8,1e-9,1e-10):
  for nref in (0,2,4):
    O.FIT_MU_REL=mu; O.FIT_REFINE=nref
    worst=0; above=0; flips=0; wd=[]; its=[]
    for (A,b,lo,hi,Js,J0),(Ap,bp) in zip(stacks,pert):
        st=[]
        w=O.critic_fit_single(A,b,np.ones(A.shape[1]),lo,hi,stats=st); its+=st
        wp=O.critic_fit_single(Ap,bp,np.ones(A.shape[1]),lo,hi)
        d=(jc(A,b,w)-Js)/max(J0,1e-12); worst=max(worst,d); above+= jc(A,b,w)>Js*(1+1e-6)+1e-6*J0+1e-12
        e=np.max(np.abs(w-wp)/np.maximum(np.abs(w),1.0)); wd.append(e); flips+= e>1e-3
    wd=np.array(wd)
    print(f"mu {mu:g} refine {nref}: worst d {worst:.3g}; above(1e-6) {above}/{len(stacks)}; perturbation 1e-14 -> |dw| median {np.median(wd):.2g} p99 {np.quantile(wd,0.99):.2g} max {wd.max():.2g}; flips(>1e-3) {flips}; iters mean {np.mean(its):.2f}",flush=True)
