"""RCCL on the one-GPU box with a process group of ONE rank: the collective calls of the sharded path (side-stream all_gather
of the theta rows through sharding.ResultGather, all_reduce MAX / MIN, barrier, the flat gradient all_reduce) execute
against the real RCCL build of this image.  What it cannot show: more than one rank (no multi-GPU node)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from sfh_amd import sharding
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = sharding.ResultGather(1, 16, dev, depth=2)
# world 1 short-cuts the gather inside ResultGather; call the collective itself too
rows = torch.arange(160.0, device=dev).reshape(16, 10)
bufs = [torch.empty_like(rows)]
side = torch.cuda.Stream(dev)
ev = torch.cuda.Event(); ev.record()
with torch.cuda.stream(side):
    side.wait_event(ev)
    dist.all_gather(bufs, rows)
    done = torch.cuda.Event(); done.record()
torch.cuda.current_stream().wait_event(done)
assert torch.equal(bufs[0], rows)
ok = torch.tensor([1], device=dev); dist.all_reduce(ok, op=dist.ReduceOp.MIN)
el = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(el, op=dist.ReduceOp.MAX)
flat, views = sharding.flat_views([(64, 3, 3, 3), (64,)], dev)
views[0].fill_(2.0)
s = sharding.allreduce_gradients(flat)
dist.barrier()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(100):
    dist.all_gather(bufs, rows)
torch.cuda.synchronize()
print("RCCL (1 rank): all_gather on a side stream, all_reduce MIN (int64) / MAX (float64), gradient all_reduce scale", s,
      ", barrier: ok;  all_gather of 640 B: %.1f us per call" % ((time.perf_counter() - t) * 1e4), "| backend", dist.get_backend())
dist.destroy_process_group()
