import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from knn_for_homology_amd import faiss, _lib
from knn_for_homology_amd.sharded import ShardedFlatIndex
L=_lib.lib(); dev=torch.device("cuda:0"); torch.cuda.set_device(0)
g=torch.Generator(device=dev); g.manual_seed(7)
d=1024
def build(nb):
    idx=ShardedFlatIndex(d,0,rank=0,world=1,row_offset=0); idx.reserve(nb)
    for i0 in range(0,nb,500000):
        m=min(500000,nb-i0); x=torch.randn((m,d),generator=g,device=dev); _lib.check(L.knn_normalize_l2_dev(x.data_ptr(),m,d,None)); idx.add_dev(x); del x
    torch.cuda.synchronize(); return idx
def timed(idx,q,k):
    for _ in range(2): idx.backend._turn=0; idx.submit(q,k)
    torch.cuda.synchronize(); best=None
    for rep in range(3):
        t0=time.perf_counter()
        for _ in range(4): idx.backend._turn=0; p=idx.submit(q,k)
        torch.cuda.synchronize(); t=(time.perf_counter()-t0)/4; best=t if best is None else min(best,t)
    p.result(); return best
for nb,nq in ((2_000_000,32),(2_000_000,1024),(14433,14433)):
    idx=build(nb); q=torch.randn((nq,d),generator=g,device=dev)
    for k in (800,1000,1100,1200,1300,1400,1536):
        print(nb,nq,k,round(1e3*timed(idx,q,k),3),flush=True)
    del idx; torch.cuda.empty_cache(); L.knn_trim()
