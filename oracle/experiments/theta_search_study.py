"""TEST INFRASTRUCTURE / study (round 6): how often alternative theta searches for CtrlNominal3WRobot land on the
minimiser the reference's trust-constr run (controllers.py:1625-1634) found on the F10 states: the build's grid walk, finer grids, a
trust-region Newton from theta = 0 with finite-difference derivatives.   python oracle/experiments/theta_search_study.py
(output kept as profiles/r06_theta_search_study.txt)"""
import numpy as np, sys
sys.path.insert(0,'.')
from oracle import nominal_oracle as NO
z=np.load('./tests/golden/F10_nominal_3wrobot.npz')
x=z['state']; xNI,eta=NO.cart2nh(x); tr=z['theta_star'].reshape(-1)
def F(i,th):
    with np.errstate(all='ignore'):
        v=NO.Fc(xNI[i:i+1],eta[i:i+1],np.array([th]))[0]
    return v if np.isfinite(v) else np.inf
def match(th): 
    d=np.abs(np.angle(np.exp(1j*(th-tr)))); return d<1e-3
# candidate A: current rule
thA=NO.theta_star(xNI,eta)
print('current', match(thA).mean())
# candidate B: trust-region Newton with FD derivatives from 0, radius 1, then golden polish in the basin found
def tr_newton(i, radius=1.0, iters=50):
    th=0.0; f=F(i,th); h=1e-4
    for _ in range(iters):
        fp,fm=F(i,th+h),F(i,th-h)
        g=(fp-fm)/(2*h); H=(fp-2*f+fm)/(h*h)
        if H>0: s=-g/H
        else: s=-np.sign(g)*radius
        s=np.clip(s,-radius,radius)
        if abs(s)<1e-9: break
        fn=F(i,th+s)
        pred=-(g*s+0.5*H*s*s) if H>0 else abs(g*s)
        rho=(f-fn)/pred if pred>0 else -1
        if rho<0.25: radius=0.25*abs(s)
        elif rho>0.75 and abs(abs(s)-radius)<1e-12: radius=min(2*radius,np.pi)
        if rho>0 and fn<f: th,f=th+s,fn
        if radius<1e-8: break
    return th
thB=np.array([tr_newton(i) for i in range(len(x))])
print('tr-newton', match(thB).mean(), np.where(~match(thB))[0])
# candidate C: finer grid walk (256 points) downhill from 0 then golden
def walk(i,n):
    h=2*np.pi/n; j=0; f=F(i,0.0)
    for _ in range(n):
        fl,fr=F(i,(j-1)*h),F(i,(j+1)*h)
        if fl<f and fl<=fr: j-=1; f=fl
        elif fr<f: j+=1; f=fr
        else: break
    a,b=(j-1)*h,(j+1)*h; g=0.6180339887498949
    for _ in range(50):
        x1=b-g*(b-a); x2=a+g*(b-a)
        if F(i,x1)<=F(i,x2): b=x2
        else: a=x1
    t=0.5*(a+b); return (t+np.pi)%(2*np.pi)-np.pi
for n in (64,128,256,512):
    thC=np.array([walk(i,n) for i in range(len(x))]); print('walk',n, match(thC).mean(), np.where(~match(thC))[0])


# ---- derivative-free candidates (comparisons of Fc values only: reproducible between the kernel and the oracle) ----
def compass(i, s0, shrink=0.5, tol=1e-9):
    th, f, s = 0.0, F(i, 0.0), s0
    while s > tol:
        fl, fr = F(i, th - s), F(i, th + s)
        if fl < f and fl <= fr:
            th, f = th - s, fl
        elif fr < f:
            th, f = th + s, fr
        else:
            s *= shrink
    return (th + np.pi) % (2 * np.pi) - np.pi


for s0 in (1.0, 0.5, 0.25, 0.1):
    thD = np.array([compass(i, s0) for i in range(len(x))])
    print('compass search from 0, first step', s0, match(thD).mean(), np.where(~match(thD))[0])


def expand_then_golden(i, d0=1e-3, grow=2.0):
    """downhill direction from the sign of Fc(+d0) - Fc(-d0), steps growing by `grow` until Fc rises, golden section on the bracket"""
    f0 = F(i, 0.0)
    sgn = -1.0 if F(i, -d0) < F(i, d0) else 1.0
    a, b, fb, d = 0.0, sgn * d0, F(i, sgn * d0), d0
    if fb >= f0:
        lo, hi = -d0, d0
    else:
        while True:
            d *= grow
            c = b + sgn * d
            fc = F(i, c)
            if fc >= fb or abs(c) > np.pi:
                lo, hi = (a, c) if sgn > 0 else (c, a)
                break
            a, b, fb = b, c, fc
    g = 0.6180339887498949
    for _ in range(60):
        x1 = hi - g * (hi - lo); x2 = lo + g * (hi - lo)
        if F(i, x1) <= F(i, x2): hi = x2
        else: lo = x1
    t = 0.5 * (lo + hi)
    return (t + np.pi) % (2 * np.pi) - np.pi


for grow in (1.5, 2.0, 3.0):
    thE = np.array([expand_then_golden(i, grow=grow) for i in range(len(x))])
    print('expanding bracket from 0 + golden section, growth', grow, match(thE).mean(), np.where(~match(thE))[0])
