import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
import sys; sys.path[:0] = [".", "fp8-quantization_amd"]
from fp8q import dist as fd
mn, mx = torch.tensor([-1.0], device=dev), torch.tensor([2.0], device=dev)
print("rccl ok", t.item(), fd.allreduce_ranges(mn, mx))
dist.destroy_process_group()
