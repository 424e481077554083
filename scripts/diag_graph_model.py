"""Round-4 diagnostic: per-step loss / loss_kd of the bench loop for a given model configuration, step graphs on vs off.
usage: python scripts/diag_graph_model.py <graph 0|1> <steps>[,<steps of a second epoch>...] -- <bench.py args>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
graph, steps = int(sys.argv[1]), [int(x) for x in sys.argv[2].split(",")]
argv = sys.argv[sys.argv.index("--") + 1:]
import torch
import bench
from moma_amd.train_student_moma import build_training
from moma_amd.learning.contrast_trainer import ContrastTrainer
from moma_amd.helper.loops_moma import train_distill_moma
from moma_amd.dataset.synthetic import SyntheticLoader
sys.argv = ["bench.py"] + argv + ([] if graph else ["--no_graph_student"])
a = bench.parse()
dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = False
opt = bench.make_opt(a, 0, 1)
opt.trace = []
torch.manual_seed(12345)
model_s, model_t, module_list, criterion_list, _tr, contrast, optimizer = build_training(opt, dev)
trainer = ContrastTrainer(opt)
_sd = os.environ.get("DIAG_SDPA")
if _sd:
    torch.backends.cuda.enable_flash_sdp(_sd == "flash"); torch.backends.cuda.enable_mem_efficient_sdp(_sd == "mem")
    torch.backends.cuda.enable_math_sdp(_sd == "math")
if os.environ.get("DIAG_FINITE", "0") == "1":
    # after every replayed step: which of the step's static tensors / gradients / parameters hold a non-finite value
    from moma_amd.helper import step_graph as _sg
    _orig = _sg.StepGraphs._replay

    def _replay(self, cap, images, labels):
        out = _orig(self, cap, images, labels)
        torch.cuda.synchronize()
        if os.environ.get("DIAG_SYNC_ONLY") == "1":
            return out
        bad = [n for n, t in list(cap.fw.items()) + [("loss_rows", cap.k2.loss_rows), ("dq", cap.k2.dq), ("images", cap.images),
                                                      ("logit_t", cap.teacher_out[0])] if torch.is_tensor(t) and not torch.isfinite(t).all()]
        ng = sum(1 for p, g in cap.grads if g is not None and not torch.isfinite(g).all())
        names = {id(p): n for m, tag in ((model_s, "s."), (criterion_list[2], "kd.")) for n, p in m.named_parameters(prefix=tag[:-1])}
        if ng and self.replays <= 3:
            for p, g in cap.grads:
                if g is not None and not torch.isfinite(g).all():
                    print("   bad grad", names.get(id(p)), tuple(g.shape), "non-finite", int((~torch.isfinite(g)).sum()), "of", g.numel())
        np_ = sum(1 for p, g in cap.grads if not torch.isfinite(p).all())
        nt = sum(1 for p in model_t.parameters() if not torch.isfinite(p).all())
        print(f"replay {self.replays}: non-finite {bad} grads {ng} params {np_} teacher params {nt} perm ok "
              f"{bool((cap.perm.static.sort().values == torch.arange(cap.perm.n, device=dev)).all())}", flush=True)
        return out
    _sg.StepGraphs._replay = _replay
for ep, n in enumerate(steps):
    loader = SyntheticLoader(n, a.batch_size, a.image_size, a.n_cls, 12345 + ep, dev)
    train_distill_moma(ep, loader, module_list, criterion_list, trainer, contrast, optimizer, opt)
torch.cuda.synchronize()
print("graphs" if graph else "eager ", "loss   ", " ".join("%.4f" % float(t[0]) for t in opt.trace))
print("graphs" if graph else "eager ", "loss_kd", " ".join("%.4f" % float(t[2]) for t in opt.trace))
