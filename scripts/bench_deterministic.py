"""bench.py with MIOpen's deterministic algorithms (torch.backends.cudnn.deterministic = True): what the reference's default
(--seed set -> cudnn.deterministic, train_student_moma.py:241-246) costs on this backend.  usage: python scripts/bench_deterministic.py [bench args]"""
import os, runpy, sys
import torch
torch.backends.cudnn.deterministic = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
