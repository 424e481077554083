"""Dual-queue memories (MoCoST: 2 terms, MoCoSSTT: 4 terms): all InfoNCE terms in one sweep (moma_infonce_fused_multi) against one
moma_infonce_fused call per term, at the benchmark shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd import ops
B, d, K, T = 256, 512, 65536, 0.15
torch.manual_seed(0)
nrm = torch.nn.functional.normalize
q, qt = nrm(torch.randn(B, d, device="cuda")), nrm(torch.randn(B, d, device="cuda"))
k, kt = nrm(q + 0.3 * torch.randn(B, d, device="cuda")), nrm(qt + 0.3 * torch.randn(B, d, device="cuda"))
ms_, mt_ = [nrm(torch.randn(K, d, device="cuda")).to(torch.bfloat16) for _ in range(2)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, terms in (("MoCoST (2 terms)", [(q, k, ms_), (q, kt, mt_)]),
                    ("MoCoSSTT (4 terms)", [(q, k, ms_), (q, kt, mt_), (qt, k, ms_), (qt, kt, mt_)])):
    for grad in (True, False):
        ts = [(a.clone().requires_grad_(grad), b, c) for a, b, c in terms]
        one = timeit(lambda: ops.infonce_fused_multi(ts, T, "bf16"))
        each = timeit(lambda: [ops.infonce_fused(a, b, c, T, "bf16") for a, b, c in ts])
        print(f"{name} dq={grad}: one sweep {one:.0f} us, one call per term {each:.0f} us", flush=True)
