"""Process-group setup and learning-rate schedules (reference: learning/base_trainer.py:13-92).

One process per GPU; `dist_backend='nccl'` is RCCL on ROCm (collectives over xGMI).  Rendezvous is read
from the environment when launched by torchrun (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), otherwise from
opt.dist_url as in the reference."""
import math
import os

import numpy as np
import torch
import torch.distributed as dist


class BaseTrainer(object):
    def __init__(self, args):
        self.args = args
        self.local_group = None
        self.logger = None

    def init_ddp_environment(self, gpu, ngpus_per_node):
        """gpu: local device index of this process; ngpus_per_node: processes on this node (reference :21-61)."""
        a = self.args
        a.gpu = gpu
        a.ngpus_per_node = ngpus_per_node
        a.node_rank = a.rank
        a.local_rank = gpu
        a.local_center = a.rank * ngpus_per_node
        if torch.cuda.is_available():
            torch.cuda.set_device(gpu)
            # MIOpen find mode on ROCm (reference :34); the CLI's --miopen_find off keeps immediate mode
            torch.backends.cudnn.benchmark = getattr(a, "miopen_find", "on") == "on"
        if a.gpu is not None:
            print("Use GPU: {} for training".format(a.gpu))
        if a.multiprocessing_distributed:
            a.rank = a.rank * ngpus_per_node + gpu
            os.environ["PYTHONWARNINGS"] = "ignore:semaphore_tracker:UserWarning"
            if not dist.is_initialized():
                if "RANK" in os.environ and "MASTER_ADDR" in os.environ:
                    dist.init_process_group(backend=a.dist_backend)
                else:
                    dist.init_process_group(backend=a.dist_backend, init_method=a.dist_url,
                                            world_size=a.world_size, rank=a.rank)
        if not dist.is_initialized():          # single process, no process group: nothing to set up
            self.local_group = None
            return
        # one group per node, used by Shuffle-BN in the reference-faithful gather mode
        groups = []
        for i in range(0, a.world_size // ngpus_per_node):
            groups.append(dist.new_group(ranks=list(range(i * ngpus_per_node, (i + 1) * ngpus_per_node)),
                                         backend=a.dist_backend))
        self.local_group = groups[a.rank // ngpus_per_node]
        if a.local_rank == 0:
            print("node_rank:", a.node_rank)
            print("local_center:", a.local_center)
            print("local group size:", dist.get_world_size(self.local_group))

    def init_tensorboard_logger(self):
        a = self.args
        if a.rank == 0:
            try:
                import tensorboard_logger as tb_logger
                self.logger = tb_logger.Logger(logdir=a.tb_folder, flush_secs=2)
            except ImportError:          # optional dependency
                self.logger = None

    def adjust_learning_rate(self, optimizer, epoch):
        a = self.args
        lr = a.learning_rate
        if a.cosine:
            eta_min = lr * (a.lr_decay_rate ** 3)
            lr = eta_min + (lr - eta_min) * (1 + math.cos(math.pi * epoch / a.epochs)) / 2
        else:
            steps = np.sum(epoch > np.asarray(a.lr_decay_epochs))
            if steps > 0:
                lr = lr * (a.lr_decay_rate ** steps)
        for g in optimizer.param_groups:
            g["lr"] = lr

    def warmup_learning_rate(self, epoch, batch_id, total_batches, optimizer):
        a = self.args
        if a.warm and epoch <= a.warm_epochs:
            p = (batch_id + (epoch - 1) * total_batches) / (a.warm_epochs * total_batches)
            lr = a.warmup_from + p * (a.warmup_to - a.warmup_from)
            for g in optimizer.param_groups:
                g["lr"] = lr
