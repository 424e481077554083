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
buf = (C.c_ulonglong * 4096)()
assert lib.moma_debug_read_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 4).astype(np.float64)
print("cycles per wave: prologue (entry->loop) %.0f   loop %.0f   [score %.0f  pv %.0f per tile]" % (a[:,2].mean(), a[:,3].mean(), a[:,0].mean()/16, a[:,1].mean()/16))
