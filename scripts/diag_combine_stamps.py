"""Timeline of infonce_combine_kernel from wall-clock stamps (needs the diagnostic build: scripts/build_k2_variants.py combine_stamps;
MOMA_HIP_LIB=moma_amd/lib/variants/libmoma_combine_stamps.so).  usage: python scripts/diag_combine_stamps.py [B] [d] [K]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moma_amd import ops, _lib
B, d, K = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 256), (2, 512), (3, 65536)))
torch.manual_seed(0)
q = torch.nn.functional.normalize(torch.randn(B, d, device="cuda")).requires_grad_(True)
k = torch.nn.functional.normalize(q.detach() + 0.3 * torch.randn(B, d, device="cuda"))
queue = torch.nn.functional.normalize(torch.randn(K, d, device="cuda")).to(torch.bfloat16)
for _ in range(6):
    ops.infonce_fused(q, k, queue, 0.15, "bf16")
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.moma_debug_combine_stamps.argtypes = [ctypes.c_void_p]
assert lib.moma_debug_combine_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
raw = buf.reshape(4096, 8)
live = raw[:, 0] != 0
st = raw[live].astype(np.int64)
n = int((st != 0).sum(1).max())
us = (st[:, :n] - st[:, 0].min()) / 100.0
names = ["entry", "row statistics done", "after the barrier", "partials of tile 0 summed", "tile 0 stored"] + [f"stamp {i}" for i in range(5, n)]
if os.environ.get("STAMPS2") == "1":
    names = ["entry", "m / x / l partials landed", "q . k rows landed + dotted", "row statistics done", "after the barrier", "partials of tile 0 summed", "tile 0 stored"] + [f"stamp {i}" for i in range(7, n)]
print(f"(B, d, K) = ({B}, {d}, {K}): {live.sum()} workgroups with stamps; us since the first workgroup's entry")
print("phase                          median    min    max")
for i in range(n):
    print(f"{names[i]:30s} {np.median(us[:, i]):6.2f} {us[:, i].min():6.2f} {us[:, i].max():6.2f}")
dur = us[:, n - 1] - us[:, 0]
print(f"per-workgroup life: median {np.median(dur):.2f} us (min {dur.min():.2f}, max {dur.max():.2f})")
if n >= 5 and os.environ.get("STAMPS2") != "1":
    print(f"phases (median): statistics {np.median(us[:, 1] - us[:, 0]):.2f}, barrier {np.median(us[:, 2] - us[:, 1]):.2f}, "
          f"tile loads + sums {np.median(us[:, 3] - us[:, 2]):.2f}, reduce + store {np.median(us[:, 4] - us[:, 3]):.2f}")
