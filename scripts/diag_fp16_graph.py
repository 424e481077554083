"""fp16 + GradScaler: gradients of the graph-served step against the eager step on IDENTICAL weights (lr = 0), per step.
usage: python scripts/diag_fp16_graph.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_step_graph as T

def run(graph):
    rec = []
    orig = torch.amp.GradScaler.step
    def step(self, optimizer, *a, **k):
        s = float(self.get_scale())
        gs = [p.grad for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
        rec.append((len(gs), float(torch.sqrt(sum((g.float() ** 2).sum() for g in gs))) / s,
                    float(gs[0].float().abs().sum()) / s, float(gs[-1].float().abs().sum()) / s))
        ps = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
        before = [p.detach().clone() for p in ps]
        mom = [optimizer.state[p]["momentum_buffer"].float().norm().item() for p in ps[:1]]
        out = orig(self, optimizer, *a, **k)
        dw = float(torch.sqrt(sum(((p.detach() - b).float() ** 2).sum() for p, b in zip(ps, before))))
        gn = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in ps)))
        rec[-1] = rec[-1] + (dw, gn, mom[0])
        return out
    torch.amp.GradScaler.step = step
    try:
        r = T._run(graph, "resnet8", True, "bf16", "bf16", "fp16", scale0=1024.0, lr=float(os.environ.get("LR", "0")))
    finally:
        torch.amp.GradScaler.step = orig
    return r, rec

a, ra = run(True)
b, rb = run(False)
print("replays", a["replays"], b["replays"])
for i, (x, y) in enumerate(zip(ra, rb)):
    print(i, "graph n=%d norm=%.5f first=%.5f last=%.5f dw=%.3e grad_after=%.4f mom0=%.4f | eager n=%d norm=%.5f first=%.5f last=%.5f dw=%.3e grad_after=%.4f mom0=%.4f | loss %.5f %.5f" % (*x, *y, a["loss"][i], b["loss"][i]))
