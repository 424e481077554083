"""Timeline of infonce_flash_kernel from wall-clock stamps (needs the diagnostic build: scripts/diag_k2_stamps.patch applied to
moma_amd/csrc/infonce_fused.hip).  usage: python scripts/diag_k2_stamps.py [B] [d] [K]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from moma_amd import ops, _lib
B, d, K = (int(sys.argv[i]) if len(sys.argv) > i else v for i, v in ((1, 256), (2, 512), (3, 65536)))
dev = "cuda"
torch.manual_seed(0)
q = torch.nn.functional.normalize(torch.randn(B, d, device=dev)).requires_grad_(True)
k = torch.nn.functional.normalize(q.detach() + 0.3 * torch.randn(B, d, device=dev))
queue = torch.nn.functional.normalize(torch.randn(K, d, device=dev)).to(torch.bfloat16)
for _ in range(6):
    ops.infonce_fused(q, k, queue, 0.15, "bf16")
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(512 * 32, dtype=np.uint64)
lib.moma_debug_k2_stamps.argtypes = [ctypes.c_void_p]
rc = lib.moma_debug_k2_stamps(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
nwg = 256 if B > 128 else 256
raw = buf.reshape(512, 32)[:nwg]
cyc = (raw >> np.uint64(32)).astype(np.int64)          # low 32 bits of s_memtime (shader cycles)
st = (raw & np.uint64(0xffffffff)).astype(np.int64)     # low 32 bits of s_memrealtime (10 ns)
n = 6 + (((K + 31) // 32 + 127) // 128 - 1 if B > 128 else ((K + 31) // 32 + 255) // 256 - 1)      # stamps per workgroup: 3 + (tiles per chunk - 1) + 3
st = st[:, :n]
t0 = st[:, 0].min()
us = (st - t0) / 100.0
names = ["entry", "Q+tile0 landed", "tile0 scores+softmax"] + [f"loop it {i}" for i in range(n - 6)] + ["last P.K, stores issued", "stores drained", "end"]
print("phase                      median    min    max   (us since the first workgroup's entry; wave 0 of 256 workgroups)")
for i in range(n):
    print(f"{names[i]:26s} {np.median(us[:, i]):6.2f} {us[:, i].min():6.2f} {us[:, i].max():6.2f}")
# in-kernel clock over the loop (first loop stamp .. last loop stamp): shader cycles per 10 ns tick
cy = ((cyc[:, n - 4] - cyc[:, 3]) & 0xffffffff).astype(np.float64)
tk = ((st[:, n - 4] - st[:, 3]) & 0xffffffff).astype(np.float64)
print(f"in-kernel clock over the loop: median {np.median(cy / tk) / 10:.3f} GHz (min {np.min(cy / tk) / 10:.3f}, max {np.max(cy / tk) / 10:.3f}); "
      f"cycles per loop iteration: median {np.median(cy) / (n - 7):.0f}")
