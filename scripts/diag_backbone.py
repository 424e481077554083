"""Diagnostic: where does first-step time go for the EffNet-B0 backbone on a fresh box (MIOpen JIT)?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moma_amd.backbones import model_dict

def log(*a):
    print(*a, flush=True)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for amp, cl in [(sys.argv[2] if len(sys.argv) > 2 else "bf16", (sys.argv[3] if len(sys.argv) > 3 else "cl") == "cl")]:
    torch.manual_seed(0)
    m = model_dict["effiB0"](num_classes=4).cuda()
    if cl:
        m = m.to(memory_format=torch.channels_last)
    x = torch.randn(B, 3, 224, 224, device="cuda")
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    opt = torch.optim.SGD(m.parameters(), lr=0.01)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(amp)
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.time()
        with torch.autocast("cuda", dtype=dt, enabled=dt is not None):
            f, o = m(x, is_feat=True)
        loss = o.float().sum()
        torch.cuda.synchronize(); t1 = time.time()
        loss.backward()
        opt.step(); opt.zero_grad()
        torch.cuda.synchronize(); t2 = time.time()
        log(f"amp={amp} cl={cl} B={B} it={it} fwd={t1-t0:.3f}s bwd+step={t2-t1:.3f}s")
