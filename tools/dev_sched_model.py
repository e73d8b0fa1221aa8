"""Dev model (round 5) of the grouped receiver's schedule at cfg 3, 8 ranks, 55 GB/s per link: a send phase (KRN-1 + first pass per slice), a FIFO wire (group 0's share of
every slice right behind that slice, then the groups one after the other) and the chain of the groups' receiver kernels. Reproduces the rehearsed 48.0 ms; every
variant of slices / group sizes stays within 0.9 ms of it (DESIGN_HISTORY.md §5.8)."""
import itertools
def sim(slices, groups, W=30.8, send=15.9, comp=28.0, fixed=0.6, early=None):
    # slices: fractions (sum 1) of reads; groups: fractions of the rank's words; early: list of (slice, group) sent right after that slice besides g0
    t_slice=[]; t=0
    for s in slices: t+=send*s; t_slice.append(t)
    q=[]  # (available_time, duration, slice, group) in enqueue order
    ns=len(slices); ng=len(groups)
    order=[]
    for s in range(ns):
        order.append((t_slice[s], s, 0))
        if early:
            for (es,eg) in early:
                if es==s: order.append((t_slice[s], s, eg))
    sent=set((s,g) for _,s,g in order)
    for g in range(1,ng):
        for s in range(ns):
            if (s,g) not in sent: order.append((t_slice[-1], s, g))
    wt=0; done={}
    for avail,s,g in order:
        start=max(wt,avail); wt=start+W*slices[s]*groups[g]; done[(s,g)]=wt
    ready=[max(done[(s,g)] for s in range(ns)) for g in range(ng)]
    ct=t_slice[-1]
    for g in range(ng):
        ct=max(ct,ready[g])+comp*groups[g]+fixed
    return ct, ready
base=sim([.5,.3,.2],[.25]*4)
print("base", base)
best=[]
for g0 in [0.25,0.3,0.35,0.4,0.45,0.5]:
    for ng in (3,4,5,6):
        rest=(1-g0)/(ng-1)
        for sl in ([.5,.3,.2],[.45,.3,.17,.08],[.6,.4],[.4,.3,.2,.1]):
            r=sim(sl,[g0]+[rest]*(ng-1))
            best.append((r[0],g0,ng,sl))
best.sort(key=lambda x:x[0])
for b in best[:8]: print(b)
# decreasing groups
for gs in ([.4,.3,.2,.1],[.35,.3,.2,.15],[.4,.25,.2,.15],[.45,.25,.18,.12],[.5,.25,.15,.1]):
    print(gs, sim([.5,.3,.2],gs)[0])
