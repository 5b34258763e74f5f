"""How often is the ordered pair (nearest, second nearest) -- or the nearest entry alone -- CONSTANT over a cell of the colour
cube, with strict inequalities at every one of its points (no tie anywhere in the cell)?  The fraction decides whether a
'settled pair' side table can pay for the headline palette palr(256, 7): a settled pixel would cost
~25 vector instructions, an unsettled one the full 79 plus ~8 for its compaction.  CPU only (numpy), ~25 s.
Result (round 4): 8^3 cells 14.9 % / 46.1 %, 4^3 cells 44.5 % / 73.4 % -- profiles/experiments/r04_headline_options.md."""
import numpy as np, time
pal = np.random.RandomState(7).randint(0,256,(256,3)).astype(np.int64)
rs = np.random.RandomState(1)
def frac(cell, ncells=3000):
    n = 256//cell
    ok = 0; okn=0
    g = np.arange(cell)
    off = np.stack(np.meshgrid(g,g,g,indexing='ij'),-1).reshape(-1,3)
    for _ in range(ncells):
        c = rs.randint(0,n,3)*cell
        pts = off + c
        d = ((pts[:,None,:]-pal[None,:,:])**2).sum(-1)
        idx = np.argsort(d,axis=1,kind='stable')[:,:3]
        dd = np.take_along_axis(d,idx,1)
        strict = (dd[:,0]<dd[:,1]).all() and (dd[:,1]<dd[:,2]).all()
        if strict and (idx[:,0]==idx[0,0]).all() and (idx[:,1]==idx[0,1]).all(): ok+=1
        if (dd[:,0]<dd[:,1]).all() and (idx[:,0]==idx[0,0]).all(): okn+=1
    return ok/ncells, okn/ncells
for cell in (8,4):
    t=time.time(); print("cell",cell,"settled pair / settled nearest:",frac(cell, 2000 if cell==8 else 4000), time.time()-t)
