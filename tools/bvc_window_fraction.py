"""Fraction of the (cell, direction) terms the BVC kernel issues with direction windows per group of G table rows
(DESIGN.md 3.2): default U(10, 30) degree angular spreads, K = 180.  Prints n, G, quad size, (issued, needed)."""
import numpy as np
rng=np.random.default_rng(0)
LOG2E=1.4426950408889634
K=180
ang=np.array([0.0]+[2*np.pi*i*2/360 for i in range(K-1)])
def frac(n, G, thr=-24.0, quad=4):
    mu_t=rng.uniform(0,2*np.pi,n); sg=np.deg2rad(rng.uniform(10,30,n))
    kappa=1/sg**2
    vm=LOG2E*kappa[:,None]*(np.cos(ang[None,:]-mu_t[:,None])-1)
    keep=vm>=thr
    band=np.minimum(keep.sum(1)//24,7)
    order=np.lexsort((np.mod(mu_t,2*np.pi),band))
    keep=keep[order]
    tot=0
    for g in range(0,n,G):
        u=keep[g:g+G].any(0)
        gap_len=0;run=0;gap_start=0
        for k in range(2*K):
            run=run+1 if not u[k%K] else 0
            if min(run,K)>gap_len: gap_len,gap_start=min(run,K),(k-min(run,K)+1)%K
        first=(gap_start+gap_len)%K
        k0=first//quad*quad
        length=min(K,(first-k0+(K-gap_len)+quad-1)//quad*quad)
        tot+=length*min(G,n-g)
    return tot/(n*K), keep.mean()
for n in (256,1024):
    for G in (4,2,1):
        for quad in (4,2):
            print(n,G,quad,frac(n,G,quad=quad))
    print(n,'thr-20',frac(n,4,thr=-20.0))
