"""Cycle split of the one-pass K2 kernel from the diagnostic stamp build (MOMA_HIP_LIB=.../lib_stamps.so; see
scripts/build_k2_variants.py).  Per wave: prologue (entry -> first barrier), first tile, tile loop, epilogue issue; per
tile: scores, P.K || softmax, wait + barrier.  Read the SHARES, not the length (the stamps forbid overlaps)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moma_amd import ops, _lib
B, d, K = 256, 512, 65536
q = torch.nn.functional.normalize(torch.randn(B, d, device="cuda")).requires_grad_(True)
k = torch.nn.functional.normalize(torch.randn(B, d, device="cuda"))
queue = torch.nn.functional.normalize(torch.randn(K, d, device="cuda")).to(torch.bfloat16)
for _ in range(20): ops.infonce_fused(q, k, queue, 0.15, "bf16")
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_uint * 16384)()
assert lib.moma_debug_read_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint32).reshape(1024, 16).astype(np.float64)
nt = 16
print("cycles per wave (mean over 1024 waves): prologue %.0f | first tile %.0f | loop %.0f [per tile: score %.0f  pv %.0f  wait+barrier %.0f] | epilogue issue %.0f"
      % tuple(v / 64 for v in (a[:, 3].mean(), a[:, 4].mean(), a[:, 5].mean(), a[:, 0].mean() / nt, a[:, 1].mean() / nt, a[:, 2].mean() / nt, a[:, 6].mean())))
print("   prologue min/median/max %.0f / %.0f / %.0f" % (a[:, 3].min() / 64, np.median(a[:, 3]) / 64, a[:, 3].max() / 64))
print("   prologue split: Q-load issue %.0f | ring DMA issue %.0f | setup + O zeroing until the wait %.0f | wait for Q + tile 0 %.0f"
      % tuple(a[:, i].mean() / 64 for i in (7, 8, 9, 10)))
