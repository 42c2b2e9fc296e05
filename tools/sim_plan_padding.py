#!/usr/bin/python3
"""Host-side pricing of pair-plan variants BEFORE a kernel is written (CPU only, numpy): slot efficiency (real memberships /
padded gather slots) and the plan's own cost model (steps + 0.2 per deliberate 2-way conflict, geneset.cpp choose_steps) for
   python tools/sim_plan_padding.py <sets> <gene slices> <step granularity 8|4|2> [pool]
with the tiles re-composed PER SLICE (sets sorted by their count in the slice; what set-indexed partial sums in LDS would
allow).  pool = 16: consecutive sets form a 16-lane group (no optimisation); larger: greedy choice inside a window (it makes
things worse -- the window spans different lengths -- and is kept as the record of that attempt).
DESIGN.md 4.6 quotes: 5000 2 8 16 -> 0.865 / 0.827;  5000 4 4 16 -> 0.822 / 0.775;  5000 1 8 16 -> 0.901 / 0.874."""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from plaid_amd import synth
g=20000; m=int(sys.argv[1]); nsl=int(sys.argv[2]); gran=int(sys.argv[3]); pool=int(sys.argv[4]) if len(sys.argv)>4 else 64
Gp,Gi=synth.geneset_csc(g,m)
width=(g+nsl-1)//nsl
def choose_steps(Lmax,deg,gran=8,cx=0.2):
    Dmax=deg.max()
    r=lambda x:((x+gran-1)//gran)*gran
    t_lo=max(gran,r(max(Lmax,(Dmax+1)//2))); t_hi=max(t_lo,r(Dmax))
    best=None
    for t in range(t_lo,t_hi+1,gran):
        cost=t+cx*np.maximum(0,deg-t).sum()
        if best is None or cost<best[0]: best=(cost,t)
    return best[1],best[0]
totT=0;totc=0
set_of=np.repeat(np.arange(m),np.diff(Gp))
for si in range(nsl):
    g0=si*width; gs=min(width,g-g0)
    sel=(Gi>=g0)&(Gi<g0+gs)
    # histogram per set of residues
    H=np.zeros((m,16),int)
    np.add.at(H,(set_of[sel],(Gi[sel]-g0)&15),1)
    cnt=H.sum(1)
    order=np.argsort(-cnt,kind='stable')
    # greedy grouping: walk sorted sets; build groups of 16 from a window of `pool` candidates
    remaining=list(order)
    groups=[]
    while remaining:
        window=remaining[:pool]
        grp=[window[0]]; deg=H[window[0]].copy(); cand=window[1:]
        while len(grp)<16 and cand:
            # pick candidate minimizing max slot
            vals=[( (deg+H[c]).max(), np.var(deg+H[c]), k) for k,c in enumerate(cand)]
            k=min(vals)[2]
            c=cand.pop(k); grp.append(c); deg+=H[c]
        for c in grp: remaining.remove(c)
        groups.append(grp)
    # tiles = 4 consecutive groups
    for t in range(0,len(groups),4):
        gg=groups[t:t+4]
        deg=np.concatenate([H[g_].sum(0) for g_ in gg])
        Lmax=max(cnt[c] for g_ in gg for c in g_)
        T,c=choose_steps(Lmax,deg,gran)
        totT+=T; totc+=c
print(f"m {m} slices {nsl} gran {gran} pool {pool}: eff {Gp[-1]/(totT*64):.3f} cost-eff {Gp[-1]/(totc*64):.3f}")
