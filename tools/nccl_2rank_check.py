"""Rehearsal of the sharded search over the nccl (RCCL) backend with 2 ranks on ONE GPU (both on cuda:0).
Run: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/nccl_2rank_check.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=rank, world_size=world)
from mmr_amd import search, synth
from oracle import search_ref

gal = synth.synth_unit_rows(20000, 512, seed=31).bfloat16()
bounds = np.linspace(0, 20000, world + 1).astype(int)
local = gal[bounds[rank]:bounds[rank + 1]].to(dev)
q = synth.synth_unit_rows(16, 512, seed=32).bfloat16()
index = search.ShardedGalleryIndex(local)
score, idx = index.search(q.to(dev), 10, 100.0)
oi, os_, _ = search_ref.cosine_topk(q, gal, 10, 100.0)
ok = np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(score.cpu().numpy(), os_)
print(f"rank {rank}: sharded search over nccl == oracle: {ok}", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
