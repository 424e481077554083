"""Which part of the step goes non-finite in the bench configuration (diagnostic)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moma_amd.miopen_env import use_shipped_db
use_shipped_db(tag="0")
import torch
import bench
from moma_amd import ops
from moma_amd.train_student_moma import build_training
from moma_amd.learning.contrast_trainer import ContrastTrainer
from moma_amd.helper.loops_moma import train_distill_moma
from moma_amd.dataset.synthetic import SyntheticLoader

variant = sys.argv[1] if len(sys.argv) > 1 else "default"
sys.argv = sys.argv[:1] + os.environ.get("DIAG_ARGS", "").split()
a = bench.parse()
dev = torch.device("cuda", 0)
opt = bench.make_opt(a, 0, 1)
torch.manual_seed(12345)
model_s, model_t, module_list, criterion_list, _tr, contrast, optimizer = build_training(opt, dev)
trainer = ContrastTrainer(opt)
kd = criterion_list[2]
if variant == "noqpack":
    contrast.qpack = lambda *a, **k: None
if variant == "nogroup":
    from moma_amd.MoMA.criterion_moco_att import Attention
    Attention.forward_group = staticmethod(lambda mods, xs: [m(x) for m, x in zip(mods, xs)])
if variant == "nooverlap":
    opt.overlap_teacher = False
opt.trace = []
for ep, n in ((0, 5), (1, int(os.environ.get("DIAG_STEPS", 6)))):
    loader = SyntheticLoader(n, a.batch_size, a.image_size, a.n_cls, 12345, dev)
    train_distill_moma(ep, loader, module_list, criterion_list, trainer, contrast, optimizer, opt)
    torch.cuda.synchronize()
    for i, (loss, idx, lkd) in enumerate(opt.trace):
        print(variant, "epoch", ep, "step", i, "loss %.5f" % float(loss), "loss_kd %.5f" % float(lkd), "index", idx, flush=True)
    opt.trace.clear()
    if variant.startswith("prime") and ep == 0:
        x0 = next(iter(loader))[0]
        teacher = trainer._graphed_teacher
        model_t.eval()
        with torch.cuda.stream(trainer._side_stream if variant != "prime_main" else torch.cuda.current_stream()), torch.autocast("cuda", dtype=torch.bfloat16):
            print("primed", teacher.prime(x0, is_feat=True))
            torch.cuda.synchronize()
            if variant == "prime_check":
                with torch.no_grad():
                    a1 = teacher(x0, is_feat=True)
                    a2 = model_t(x0, is_feat=True)
                print("replay vs eager logits", a1[1].float().abs().max().item(), a2[1].float().abs().max().item(), (a1[1].float() - a2[1].float()).abs().max().item())
                print("finite", torch.isfinite(a1[1]).all().item(), torch.isfinite(a1[0][-1]).all().item())
    for nm, mod in (("student", model_s), ("teacher", model_t), ("atts_q", kd.atts_q), ("atts_k", kd.atts_k), ("embed_s", kd.embed_s), ("embed_t", kd.embed_t)):
        bad = [k for k, v in mod.state_dict().items() if v.is_floating_point() and not torch.isfinite(v).all()]
        print("  ", nm, "non-finite tensors:", bad[:4], flush=True)
    print("   queue finite:", bool(torch.isfinite(contrast.memory.float()).all()), flush=True)
