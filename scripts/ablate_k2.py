"""Time the one-pass K2 kernel alone (library-recorded events) for one libmoma_hip build (MOMA_HIP_LIB)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd import ops
B, d, K = 256, 512, 65536
torch.manual_seed(0)
q = torch.nn.functional.normalize(torch.randn(B, d, device="cuda"))
k = torch.nn.functional.normalize(q + 0.3 * torch.randn(B, d, device="cuda"))
queue = torch.nn.functional.normalize(torch.randn(K, d, device="cuda")).to(torch.bfloat16)
hip = C.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
pairs = []
def prov():
    a, b = C.c_void_p(), C.c_void_p()
    hip.hipEventCreate(C.byref(a)); hip.hipEventCreate(C.byref(b))
    pairs.append((a, b)); return a.value, b.value
for grad in (True, False):
    qq = q.clone().requires_grad_(grad)
    ops.set_kernel_event_provider(None)
    for _ in range(5): ops.infonce_fused(qq, k, queue, 0.15, "bf16")
    pairs.clear(); ops.set_kernel_event_provider(prov)
    for _ in range(30): ops.infonce_fused(qq, k, queue, 0.15, "bf16")
    torch.cuda.synchronize()
    t = []
    for a, b in pairs:
        ms = C.c_float(); hip.hipEventElapsedTime(C.byref(ms), a, b); t.append(ms.value)
    t.sort()
    print(f"{os.path.basename(os.environ.get('MOMA_HIP_LIB','default')):55s} dq={grad}: one-pass kernel median {t[len(t)//2]*1e3:.1f} us  min {t[0]*1e3:.1f} us", flush=True)
