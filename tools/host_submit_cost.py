#!/usr/bin/env python3
"""One GPU, a real NCCL (RCCL) group of one rank: what a row-sharded step costs the HOST (ShardedFlatIndex.submit: keys ->
all_gather_into_tensor -> merge, alternating lanes) beside its GPU time, on a tiny shard (the launch path alone) and on a
1.25 M-row shard (one of eight of the 10 M-row database).  The host must stay well under the 0.9 ms a shard step takes."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
os.environ["KNN355_FORCE_COLLECTIVE"] = "1"
import torch, torch.distributed as dist
from knn_for_homology_amd import faiss
from knn_for_homology_amd.sharded import ShardedFlatIndex
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
for nb in (50_000, 1_250_000):
    idx = ShardedFlatIndex(1024, 0, rank=0, world=1, row_offset=0)
    x = torch.randn((nb, 1024), device=dev)
    idx.add_dev(x)
    q = torch.randn((32, 1024), device=dev)
    for _ in range(20): idx.submit(q, 100)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    pend = [idx.submit(q, 100) for _ in range(K)]
    t_submit = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"nb={nb}: host submit {1e6*t_submit/K:.0f} us/step, wall {1e6*t_all/K:.0f} us/step (keys -> all_gather(nccl, world 1) -> merge, two lanes)", flush=True)
    del idx, x
dist.destroy_process_group()
