"""Worker of tests/test_torchrun_cpu.py: started by `python -m torch.distributed.run` (2 ranks, gloo, CPU), runs the
trainer's own `main()` -- torchrun branch (reference train_student_moma.py:207-224), `main_worker`,
`BaseTrainer.init_ddp_environment` (reference learning/base_trainer.py:21-61), DDP wrap, epoch loop, per-rank checkpoint.
The four kernel wrappers are replaced by the torch-CPU stand-ins of tests/test_dp_gloo.py (host logic only)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MOMA_DIST_BACKEND"] = "gloo"

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tests.test_dp_gloo import _install_cpu_standins  # noqa: E402

torch.set_num_threads(1)
_install_cpu_standins()
from moma_amd import train_student_moma as T  # noqa: E402

T._REQUIRE_GPU = False                                   # host-logic test: the kernel wrappers are stand-ins here
from moma_amd.learning.base_trainer import BaseTrainer  # noqa: E402

out_dir = sys.argv[1]
seen = {}
orig_init = BaseTrainer.init_ddp_environment


def spy(self, gpu, ngpus_per_node):                     # record what the bring-up produced on this rank
    orig_init(self, gpu, ngpus_per_node)
    a = self.args
    seen.update(rank=a.rank, node_rank=a.node_rank, local_rank=a.local_rank, world_size=a.world_size,
                ngpus_per_node=a.ngpus_per_node, backend=dist.get_backend(),
                local_group_size=dist.get_world_size(self.local_group), initialized=dist.is_initialized())


BaseTrainer.init_ddp_environment = spy
T.main(["--distill", "moma", "--model_s", "resnet8", "--model_t", "resnet8", "--dataset", "cifar100", "--n_cls", "4",
        "--batch_size", "4", "--epochs", "1", "--steps_per_epoch", "3", "--nce_k", "64", "--head", "mlp", "--feat_dim", "32",
        "-c", "1", "-d", "1", "-b", "1", "--moma_prec", "fp32", "--print_freq", "1", "--save_root", out_dir,
        "--no_graph_teacher", "--no_overlap_teacher"] + sys.argv[2:])
json.dump(seen, open(os.path.join(out_dir, f"seen_rank{os.environ['RANK']}.json"), "w"))
