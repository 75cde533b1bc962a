#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (an experiment ON the oracle, kept under oracle/ because it imports it; nothing in the product path,
tools/ or bench.py uses it).

Round 3 experiment: can the critic fit (oracle.rcg_oracle.critic_fit_single, k_critic_fit) reach its minimiser in a
BOUNDED, small number of iterations?  (block principal pivoting or projected Newton.)  CPU only.

Collects the TD stacks (A, b) that RQL / SQL closed loops of the oracle actually produce and solves each with
  walk   the single-pivot primal active-set walk of rounds 1-2 (the oracle's critic_fit_single)
  bpp    block principal pivoting (all infeasible variables change sides at once; Portugal / Judice / Vicente safeguard:
         3 block pivots without a new minimum of the infeasible count, then Murty single pivots)
  pn     projected Newton with block release and a backtracking search along the projection arc (Bertsekas)
  arc    the walk with an EXACT search along the projection arc (every variable that reaches a bound before the arc's
         minimiser is fixed in the same iteration); optionally block release
and prints, per problem family, iterations (mean, mean of the per-wave maximum = what a 64-lane wave pays, maximum) and
how many fits ended above the walk's objective.  Result on this container (256 tank / 64 robot envs x 24 ticks):

  2tank quadratic RQL (configs[2]):  walk mean 2.1 wave-max 7.7 max 16 | bpp max 28 = the cap, 29 / 3072 fits end ABOVE
      the walk (cycling on the collinear, saturated stacks of a closed loop: sigma(A) = 2.3 / 0.17 / 0.003, b ~ 400,
      solution = 4 of 6 variables on a bound) | pn max 60 = the cap, 84 worse | arc wave-max 6.2, max 13-15, 0 worse
  3wrobotNI quadratic RQL:           walk wave-max 17.6 max 39 | bpp max 55 (cap), 70 / 1536 worse | arc wave-max 41 max 120
  3wrobot quad-lin SQL (dc = 35):    walk wave-max 15.4 max 71 | bpp max 115 (cap) | arc (block release) wave-max 7.8 max 30

Reading: at the optimum of these stacks the residual is non-zero, so at most m - 1 = 2 variables are free and the rest
sit on a bound - the end point is (next to) a vertex of the box, far from the interior start w_init = ones, and every
variable that has to go from one bound to the other costs a pivoting method two iterations.  Block pivots cycle on them
(mu = 1e-8: the free-subspace Newton point is 1e4 box widths away), the monotone variants save 20 %.  A walk truncated at
c iterations would bound the count but makes the result depend on the path, i.e. on rounding (the kernel fuses its
multiply-adds, numpy does not): parity with the oracle would then hold only up to pivot ties.  k_critic_fit therefore
keeps the exact single-pivot walk; DESIGN.md records this under "critic fit".

    python oracle/experiments/fit_bpp_experiment.py [n_envs] [ticks]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rcg_oracle as O  # noqa: E402
from tests.helpers import oracle_cfg  # noqa: E402


def collect(name, cs, mode, B, T, Nh=6, K=16, seed=0):
    """Run the oracle's closed loop and return every (A, b) its fits saw."""
    rng = np.random.default_rng(seed)
    cfg = oracle_cfg(name, n_actor=Nh, mode=mode, critic_struct=cs, n_critic=4, buffer_size=10, gamma=1.0)
    x0 = np.stack([rng.uniform(0, 2, B), rng.uniform(-2, 2, B)], -1) if name == "2tank" else rng.uniform(-3, 3, (B, cfg.ds))
    env = O.new_batch(cfg, x0)
    cand = O.grid_candidates(cfg, K)
    stacks = []
    orig = O.critic_td_system

    def spy(w_prev, obs_buf, act_buf, c):
        A, b = orig(w_prev, obs_buf, act_buf, c)
        stacks.append((A.copy(), b.copy()))
        return A, b

    O.critic_td_system = spy
    try:
        for _ in range(T):
            O.control_tick(cfg, env, cand)
    finally:
        O.critic_td_system = orig
    lo, hi = O.critic_bounds(cs, cfg.dc)
    return stacks, np.ones(cfg.dc), lo, hi


def fobj(A, b, w, w0, mu):
    r = A @ w - b
    return 0.5 * r @ r + 0.5 * mu * (w - w0) @ (w - w0)


def bpp(A, b, w0, lo, hi, tries0=3):
    """Block principal pivoting with the Portugal / Judice / Vicente safeguard; returns (w, iterations)."""
    m, dc = A.shape
    mu = max(O.FIT_MU_REL * float((A * A).sum()) / m, 1e-30)
    w = np.clip(w0, lo, hi).astype(float)
    free = (w > lo) & (w < hi)
    at_hi = (~free) & (w >= hi)
    ninf, tries = dc + 1, tries0
    cap = O.fit_max_iters(dc)
    for it in range(1, cap + 1):
        F = free
        M = A[:, F] @ A[:, F].T + mu * np.eye(m)
        lam = np.linalg.solve(M, b - A[:, ~F] @ w[~F] - A[:, F] @ w0[F])
        w[F] = w0[F] + A[:, F].T @ lam
        res = A @ w - b
        g = A.T @ res + mu * (w - w0)
        scale = np.abs(A * res[:, None]).sum(0) + np.abs(mu * (w - w0))
        viol = np.where(F, (w < lo) | (w > hi), np.where(at_hi, g, -g) > O.FIT_KKT_TOL * scale)
        nv = int(viol.sum())
        if nv == 0:
            return w, it
        if nv < ninf:
            ninf, tries, block = nv, tries0, True
        elif tries > 0:
            tries, block = tries - 1, True
        else:
            block = False
        idx = np.flatnonzero(viol) if block else np.flatnonzero(viol)[-1:]
        for i in idx:
            if free[i]:
                at_hi[i] = w[i] > hi[i]
                w[i] = hi[i] if at_hi[i] else lo[i]
                free[i] = False
            else:
                free[i] = True
    w = np.clip(w, lo, hi)
    wi = np.clip(w0, lo, hi)
    return (w if fobj(A, b, w, w0, 0.0) <= fobj(A, b, wi, w0, 0.0) else wi), cap


def proj_newton(A,b,w0,lo,hi,max_it=60,trace=False,beta=0.25):
    m,dc=A.shape
    mu=max(O.FIT_MU_REL*float((A*A).sum())/m,1e-30)
    w=np.clip(w0,lo,hi).astype(float)
    f=fobj(A,b,w,w0,mu); nev=0
    for it in range(1,max_it+1):
        res=A@w-b; g=A.T@res+mu*(w-w0)
        scale=np.abs(A*res[:,None]).sum(0)+np.abs(mu*(w-w0))
        tol=1e-10*scale
        fixed=((w<=lo)&(g>tol))|((w>=hi)&(g<-tol))|((w<=lo)&(np.abs(g)<=tol))|((w>=hi)&(np.abs(g)<=tol))
        F=~fixed
        if not F.any(): return w,it
        M=A[:,F]@A[:,F].T+mu*np.eye(m)
        rhs=b-A[:,~F]@w[~F]-A[:,F]@w0[F]
        lam=np.linalg.solve(M,rhs)
        z=w.copy(); z[F]=w0[F]+A[:,F].T@lam
        d=z-w
        if np.all(np.abs(d)<=1e-12*np.maximum(np.abs(w),1)): return w,it
        a=1.0
        while True:
            wn=np.clip(w+a*d,lo,hi); fn=fobj(A,b,wn,w0,mu); nev+=1
            if fn< f - 1e-12*abs(f) or a<1e-12: break
            a*=beta
        if trace: print(f"   it {it}: |F|={F.sum()} alpha {a:.3g} f {f:.6e}->{fn:.6e} w={wn}")
        if a<1e-12: return w,it
        w,f=wn,fn
    return w,max_it


def arc_walk(A,b,w0,lo,hi,block_release=False,max_it=120,trace=False):
    m,dc=A.shape
    mu=max(O.FIT_MU_REL*float((A*A).sum())/m,1e-30)
    w=np.clip(w0,lo,hi).astype(float)
    free=(w>lo)&(w<hi); at_hi=(~free)&(w>=hi)
    nsolve=0
    for it in range(1,max_it+1):
        F=free
        M=A[:,F]@A[:,F].T+mu*np.eye(m)
        rhs=b-A[:,~F]@w[~F]-A[:,F]@w0[F]
        lam=np.linalg.solve(M,rhs); nsolve+=1
        z=w.copy(); z[F]=w0[F]+A[:,F].T@lam
        viol=F&((z<lo)|(z>hi))
        if viol.any():
            d=z-w
            # exact search along the projection arc p(a)=clip(w+a d), a in [0,1]
            bp=np.where(viol,(np.where(z<lo,lo,hi)-w)/np.where(d==0,1,d),np.inf)
            moving=F.copy(); a=0.0; p=w.copy()
            r=A@p-b
            while True:
                dS=np.where(moving,d,0.0)
                AdS=A@dS
                q1=r@AdS+mu*((p-w0)@dS); q2=AdS@AdS+mu*(dS@dS)
                nxt=np.where(moving,bp,np.inf); j=int(np.argmin(nxt)); anext=min(nxt[j],1.0)
                if q1>=0: break
                astar=a-q1/q2 if q2>0 else np.inf
                if astar<=anext:
                    p=p+(astar-a)*dS; r=r+(astar-a)*AdS; a=astar; break
                p=p+(anext-a)*dS; r=r+(anext-a)*AdS; a=anext
                if anext>=1.0: break
                # fix every variable whose breakpoint is here
                hit=moving&(bp<=a)
                p[hit]=np.where(z[hit]<lo[hit],lo[hit],hi[hit]); moving&=~hit
            hitall=F&viol&(bp<=a)
            if not hitall.any():  # (cannot happen: the first breakpoint is always reached or q1>=0 at a=0 impossible)
                j=int(np.argmin(np.where(viol,bp,np.inf))); hitall[j]=True
            w=np.clip(p,lo,hi)
            for j in np.flatnonzero(hitall):
                at_hi[j]=z[j]>hi[j]; w[j]=hi[j] if at_hi[j] else lo[j]; free[j]=False
            if trace: print(f"  it {it}: arc a={a:.3e} fixed {np.flatnonzero(hitall)} f={fobj(A,b,w,w0,mu):.6e}")
            continue
        w[F]=z[F]
        res=A@w-b
        g=A.T@res+mu*(w-w0)
        scale=np.abs(A*res[:,None]).sum(0)+np.abs(mu*(w-w0))
        score=np.where(at_hi,g,-g); score[F]=-np.inf
        cand=(~F)&(score>1e-10*scale)
        if not cand.any(): return w,it
        if block_release:
            free|=cand
            if trace: print(f"  it {it}: release {np.flatnonzero(cand)} f={fobj(A,b,w,w0,mu):.6e}")
        else:
            j=int(np.argmax(np.where(cand,score,-np.inf))); free[j]=True
            if trace: print(f"  it {it}: release {j} f={fobj(A,b,w,w0,mu):.6e}")
    return w,max_it


def wave_max(x):
    x = np.array(x)
    n = len(x) // 64 * 64
    return float(x[:n].reshape(-1, 64).max(1).mean()) if n else float(x.max())


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    inv = {v: k for k, v in O.CRITIC_IDS.items()}
    for name, cs, mode in (("2tank", O.CRITIC_QUADRATIC, O.MODE_RQL), ("2tank", O.CRITIC_QUAD_LIN, O.MODE_RQL),
                           ("3wrobotNI", O.CRITIC_QUADRATIC, O.MODE_RQL), ("3wrobot", O.CRITIC_QUAD_LIN, O.MODE_SQL)):
        stacks, w0, lo, hi = collect(name, cs, mode, B if name == "2tank" else max(64, B // 4), T)
        solvers = {"bpp": bpp, "pn": proj_newton, "arc": arc_walk,
                   "arc+block release": lambda *a: arc_walk(*a, block_release=True)}
        its = {k: [] for k in solvers}
        worse = {k: 0 for k in solvers}
        itw = []
        for A, b in stacks:
            for e in range(A.shape[0]):
                st = []
                ww = O.critic_fit_single(A[e], b[e], w0, lo, hi, stats=st)
                itw.append(st[0])
                mu = max(O.FIT_MU_REL * float((A[e] * A[e]).sum()) / A[e].shape[0], 1e-30)
                f_walk = fobj(A[e], b[e], ww, w0, mu)
                for k, fn in solvers.items():
                    w, it = fn(A[e], b[e], w0, lo, hi)
                    its[k].append(it)
                    if fobj(A[e], b[e], w, w0, mu) > f_walk * (1 + 1e-7) + 1e-12:
                        worse[k] += 1
        tag = f"{name} {inv[cs]} {'RQL' if mode == O.MODE_RQL else 'SQL'}"
        print(f"{tag}: {len(itw)} fits | walk mean {np.mean(itw):.2f} wave-max {wave_max(itw):.1f} max {np.max(itw)}")
        for k in solvers:
            print(f"    {k:18s} mean {np.mean(its[k]):.2f} wave-max {wave_max(its[k]):.1f} max {np.max(its[k])} "
                  f"| ends above the walk: {worse[k]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
