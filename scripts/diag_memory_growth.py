"""Device-memory growth of the graph-served loop over several epochs (allocated / reserved after each epoch; flat once every variant is captured).  usage: python scripts/diag_memory_growth.py [epochs] [steps] -- <bench.py args>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
epochs, steps = int(sys.argv[1]), int(sys.argv[2])
argv = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
import torch
import bench
from moma_amd.train_student_moma import build_training
from moma_amd.learning.contrast_trainer import ContrastTrainer
from moma_amd.helper.loops_moma import train_distill_moma
from moma_amd.dataset.synthetic import SyntheticLoader
sys.argv = ["bench.py"] + argv
a = bench.parse()
dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = False
opt = bench.make_opt(a, 0, 1)
torch.manual_seed(1)
model_s, model_t, module_list, criterion_list, _tr, contrast, optimizer = build_training(opt, dev)
trainer = ContrastTrainer(opt)
rows = []
for ep in range(epochs):
    loader = SyntheticLoader(steps, a.batch_size, a.image_size, a.n_cls, 100 + ep, dev)
    train_distill_moma(ep, loader, module_list, criterion_list, trainer, contrast, optimizer, opt)
    torch.cuda.synchronize()
    rows.append((torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.memory_reserved() / 2 ** 20))
    print(f"epoch {ep}: allocated {rows[-1][0]:.1f} MiB  reserved {rows[-1][1]:.1f} MiB", flush=True)
# (steps up are expected while variants are still being captured: the step graphs in epoch 0, the teacher's eval-mode forward --
#  seen once per epoch -- in epoch 3; after that the numbers must be flat)
grow = rows[-1][0] - rows[-2][0]
print("growth of allocated memory over the last epoch: %.2f MiB (reserved %.1f -> %.1f)" % (grow, rows[-2][1], rows[-1][1]))
sys.exit(0 if abs(grow) < 8 else 1)
