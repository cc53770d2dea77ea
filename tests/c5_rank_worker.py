"""One rank of test_baseline_size.py::test_config5_two_gloo_ranks_on_one_gpu (launched by torch.distributed.run).

BASELINE config 5 batch-sharded over WORLD_SIZE ranks that share cuda:0 (gloo carries the 16-byte range exchange):
every rank holds [512 / W, 4096, 512] of each of two sequential batches, folds its running min/max, all-reduces the
range, quantizes its part to E4M3 with the GLOBAL range.  Checked here against the CPU oracle: the range after every
batch equals the fold of all ranks' oracle min/max, the quantized part equals the oracle's on 64 windows."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "fp8-quantization_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    import oracle
    from fp8q import dist as fd
    from test_baseline_size import _bits_equal, _check_windows, _np, _slab, SLAB
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    shape = (SLAB[0] // world,) + SLAB[1:]
    state, g_mn, g_mx = None, None, None
    for b in range(2):
        x = _slab(1234 + 1000 * b + rank, shape)
        y, state = fd.calibrate_quantize_sharded(x, 3, 8, 1, state=state)
        mn, mx = oracle.c_minmax(_np(x).reshape(-1), False)
        parts = [None] * world
        dist.all_gather_object(parts, (mn, mx))
        for pmn, pmx in parts:                       # fold order is irrelevant: min / max are associative
            g_mn, g_mx = (pmn, pmx) if g_mn is None else oracle.c_fold(g_mn, g_mx, pmn, pmx, 1)
        _bits_equal(_np(state[0]), g_mn, f"rank {rank} batch {b}: global running min")
        _bits_equal(_np(state[1]), g_mx, f"rank {rank} batch {b}: global running max")
        _check_windows(x, y, oracle.c_absmax(g_mn, g_mx), 3, 7 * rank + b, f"rank {rank} batch {b}: E4M3 part")
        del x, y
    print(f"C5_RANK_OK rank={rank} world={world} range={float(g_mn[0])!r},{float(g_mx[0])!r}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
