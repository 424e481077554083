"""configs[4] at full size, fp16 + GradScaler: per-step scale / found-inf / gradient norm / weight delta, graph-served vs eager.
usage: python scripts/diag_cfg4_graph.py [graph|eager]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_configs_fullsize as T

mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
rec = []
where = []
orig_step, orig_update = torch.amp.GradScaler.step, torch.amp.GradScaler.update
def step(self, optimizer, *a, **k):
    ps = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    if os.environ.get("SYNC") == "before":
        torch.cuda.synchronize()
    g2 = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in ps))            # (device tensors: nothing is read back before the step)
    nf = sum((~torch.isfinite(p.grad)).sum() for p in ps)
    per = torch.stack([(~torch.isfinite(p.grad)).sum() for p in ps])
    mx = torch.stack([torch.nan_to_num(p.grad.float(), nan=0.0, posinf=0.0, neginf=0.0).abs().max() for p in ps])
    where.append((per, mx, [tuple(p.shape) for p in ps]))
    sc = self._scale.clone()
    w0 = [p.detach().clone() for p in ps[:40]]
    out = orig_step(self, optimizer, *a, **k)
    dw = torch.sqrt(sum(((p.detach() - b).float() ** 2).sum() for p, b in zip(ps[:40], w0)))
    rec.append(dict(scale=sc, n=len(ps), gnorm=g2, nonfinite=nf, dw=dw))
    return out
torch.amp.GradScaler.step = step
argv = ["--model", "ResNet50", "--model_t", "vit_base_patch16_224", "--image_size", "512", "--batch_size", "64", "--amp", "fp16",
        "--learning_rate", "2e-4", "--no_cpu_baseline"] + (["--no_graph_student"] if mode == "eager" else [])
l, k, replays, finite = T._losses(argv, [5, 6])
print(mode, "replays", replays)
for i, (per, mx, shapes) in enumerate(where):
    per = per.cpu().tolist(); mx = mx.cpu().tolist()
    bad = [(j, shapes[j], per[j]) for j in range(len(per)) if per[j]]
    top = sorted(range(len(mx)), key=lambda j: -mx[j])[:3]
    print("step", i, "non-finite in", bad[:6], "| largest finite scaled grads:", [(j, shapes[j], "%.3g" % mx[j]) for j in top])
for i, r in enumerate(rec):
    print(i, "loss %.5f kd %.5f scale %.0f n %d gnorm %.4f nonfinite %d dw %.3e" % (l[i], k[i], float(r["scale"]), r["n"], float(r["gnorm"]) / float(r["scale"]), int(r["nonfinite"]), float(r["dw"])))
