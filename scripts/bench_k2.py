"""Micro-benchmark of K2 (InfoNCE over the queue) through the C ABI: HIP-event time per launch, algorithmic
GB/s and TFLOP/s.  usage: python scripts/bench_k2.py [B] [d] [K] [queue_dtype] [prec] [iters] [dq_only]
(dq_only: only the launches with the gradient -- the training step's call -- so that per-kernel counter means of the combine
kernel, whose name does not tell the two kinds apart, are those of that call)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
qdt = sys.argv[4] if len(sys.argv) > 4 else "bf16"
prec = sys.argv[5] if len(sys.argv) > 5 else "bf16"
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 50
dq_only = len(sys.argv) > 7 and sys.argv[7] == "dq_only"
if os.environ.get("MOMA_K2_TARGET_WG"):                    # (read HERE, by the sweep script -- the library reads no environment)
    ops.debug_set_k2_target_wg(int(os.environ["MOMA_K2_TARGET_WG"]))
torch.manual_seed(0)
dev = "cuda"
q = torch.nn.functional.normalize(torch.randn(B, d, device=dev))
k = torch.nn.functional.normalize(q + 0.3 * torch.randn(B, d, device=dev))
queue = torch.nn.functional.normalize(torch.randn(K, d, device=dev)).to(torch.bfloat16 if qdt == "bf16" else torch.float32)
qbytes = queue.element_size()
for grad in ((True,) if dq_only else (True, False)):
    qq = q.clone().requires_grad_(grad)
    for _ in range(5):
        ops.infonce_fused(qq, k, queue, 0.15, prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.infonce_fused(qq, k, queue, 0.15, prec)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    byt = K * d * qbytes + (4 if grad else 3) * B * d * 4 + 12 * B
    flops = (4.0 if grad else 2.0) * B * d * (K + 1)
    print(f"K2 B={B} d={d} K={K} queue={qdt} prec={prec} dq={grad}: {ms*1e3:.1f} us/launch  "
          f"{byt/ms/1e6:.0f} GB/s ({byt/ms/1e6/8000*100:.1f}% of 8 TB/s)  {flops/ms/1e9:.0f} TFLOP/s "
          f"({flops/ms/1e9/2500*100:.1f}% of 2.5 PF)", flush=True)
