#!/usr/bin/env python3
"""Where an optimiser tick with a many-weight critic goes, tick by tick of a closed loop from random states (3-wheel robot, RQL, C2 batch,
5 optimiser iterations, 4 pairs): the fit's launch (env step + push + fit) and the optimiser's, in us, every third tick - the fit is short
while the buffers fill and as long as its active-set walk once half of the weights sit on a bound.   python tools/fit_ticks_probe.py"""
import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch
from valu_probe import states
from rcognita_amd import Engine, _native as N
from rcognita_amd.pool import preset_engine_config
B=65536
rng=np.random.default_rng(7)
for cs in ("quad-lin","quadratic","quad-mix"):
    e=Engine(preset_engine_config("3wrobot",B,Nactor=10,dtype="f32",mode="RQL",critic_struct=cs,buffer_size=10))
    e.set_state(states(rng,"3wrobot",B))
    e.set_optimizer(4)
    e.profile((N.KERNEL_CRITIC,N.KERNEL_ACTOR),stride=1)
    out=[]; pc=pa=0.0
    for t in range(46):
        e.control_tick_opt(iters=5); e.synchronize()
        cm,cn=e.profile_read(N.KERNEL_CRITIC); am,an=e.profile_read(N.KERNEL_ACTOR)
        out.append((round((cm-pc)*1e3), round((am-pa)*1e3))); pc,pa=cm,am
    w=e.get_field(N.FIELD_W_CRITIC)
    lo,hi=(-1e3,1e3) if cs in("quad-lin","quad-mix") else (0,1e3)
    print(cs, "fit/opt us per tick:", out[::3], "weights at a bound: %.1f%%"%(100*np.mean((w<=lo)|(w>=hi))), "frozen", int(np.sum(e.get_field(N.FIELD_STATUS)&1)))
    e.close()
