"""De-risk the RCCL calls of bench.py / dnastore_amd.shard on a one-GPU box: a process group of world size 1 over the
"nccl" backend (= RCCL) runs every collective the multi-GPU path uses -- barrier, all_reduce MAX/SUM on int64 and float64,
broadcast_object_list, scatter of uint8, gather of uint8 / int32 / float64 / int64 -- with the dtypes and devices it uses
them with.  (Collectives over one rank move no data; what this catches is an unsupported dtype, op or argument.)
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29520 RANK=0 WORLD_SIZE=1 python tools/rccl_world1_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
from dnastore_amd import shard

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29520")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
shard._dist_ready = lambda world: True          # run the collectives although world == 1
dist.barrier()
t = torch.tensor([7], dtype=torch.int64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert int(t) == 7
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.SUM); assert float(t[1]) == 2.5
# scatter of packed reads
off = np.array([0, 3, 7, 7], dtype=np.uint64)
bases = np.array([0, 1, 2, 3, 0, 1, 2], dtype=np.uint8)
idx, o2, d_bases = shard.scatter_reads(off, bases, 1, 0, dev)
assert list(idx) == [0, 1, 2] and d_bases.is_cuda and d_bases[:7].cpu().numpy().tolist() == bases.tolist(), (idx, o2)
# gather of results (uint8 symbols, int32 lengths, float64 log-likelihoods, uint8 status)
k, cap = 3, 5
sym = torch.arange(k * cap, dtype=torch.uint8, device=dev)
ln = torch.tensor([1, 2, 3], dtype=torch.int32, device=dev)
ll = torch.tensor([-1., -2., -3.], dtype=torch.float64, device=dev)
st = torch.zeros(k, dtype=torch.uint8, device=dev)
res = shard.gather_results(sym, ln, ll, st, 1, 0)
assert len(res) == 1 and torch.equal(res[0][0], sym) and torch.equal(res[0][1], ln) and torch.equal(res[0][2], ll)
# the E-step reduction
c, l2 = shard.allreduce_counts(np.arange(27, dtype=np.float64), -12.5, 1, dev)
assert c.tolist() == list(range(27)) and l2 == -12.5
dist.barrier()
dist.destroy_process_group()
print("RCCL world-size-1 check: ok (torch %s, nccl/rccl %s)" % (torch.__version__, ".".join(map(str, torch.cuda.nccl.version()))))
