"""`train_student_moma.py` CLI for the MoMA contrastive-distillation path on MI355X.

Keeps every flag and default of the reference's argument parser (train_student_moma.py:46-131) and the
startup order of its main_worker (:227-392: seed -> probe batch -> student, teacher -> s_dim/t_dim probe ->
build_mem -> broadcast_memory -> CMO -> SGD over the trainable list -> DDP(model_s)), so seeds, checkpoints
and launch lines carry over.  What differs:
  * only `--distill moma` (and `kd`) are built -- the other criteria are out of scope;
  * real datasets need author-local folders; `--dataset synthetic` (default when the requested dataset is not
    available) feeds pre-generated batches of the same shape;
  * one process per GPU either through torchrun (RANK/LOCAL_RANK/WORLD_SIZE in the environment) or, as in the
    reference, `--multiprocessing-distributed` + mp.spawn; the backend string 'nccl' is RCCL on ROCm;
  * new optional flags: --moma_prec, --queue_dtype, --amp, --channels_last, --shuffle_bn, --no_fused,
    --steps_per_epoch, --num_heads, --no_graph_teacher, --no_graph_student, --dp (student wrap at world size > 1: flat = one gradient all-reduce per step, the
    default; ddp = stock DistributedDataParallel as in the reference);
  * the reference's default run is cudnn.deterministic (a seed is set: :241-246; cudnn.benchmark only with its --deterministic
    switch, :417-418).  MIOpen's deterministic algorithms cost 3.9x at BASELINE configs[1] (158 vs 40 ms per step), so here that
    mode is opt-in: `--reproducible` (seeded, deterministic algorithms, no find mode).  This library adds nothing unordered (no
    atomics): inside one process the loop then repeats bit for bit (tests/test_gpu_step_graph.py); between processes MIOpen's own
    solver choice can still differ (scripts/diag_cli_trace.py).  The default is MIOpen's fast algorithms.
"""
from __future__ import print_function

import argparse
import os
import random
import time

from .miopen_env import use_shipped_db

use_shipped_db(tag=os.environ.get("LOCAL_RANK", ""))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.optim as optim

from .MoMA.mem_moco import build_mem
from .MoMA.criterion_moco_att import CMO
from .dataset.synthetic import SyntheticLoader
from .distiller_zoo import DistillKL
from .helper.loops_moma import macro_f1, train_distill_moma, validate_distill
from .helper.util import adjust_learning_rate, reduce_tensor, save_dict_to_json, update_dict_to_json
from .learning.contrast_trainer import ContrastTrainer
from .model_def import load_model
from .backbones import model_dict

N_CLS = {"cifar100": 100, "imagenet": 1000, "colon_tma_manual": 4, "panda_512": 4, "prostate_hv": 4, "gastric": 8}
IMAGE_SIZE = {"cifar100": 32, "imagenet": 224}


def build_parser():
    p = argparse.ArgumentParser("argument for training")
    # basic
    p.add_argument("--print_freq", type=int, default=50, help="print frequency")
    p.add_argument("--batch_size", type=int, default=64, help="batch_size")
    p.add_argument("--num_workers", type=int, default=8, help="num of workers to use")
    p.add_argument("--epochs", type=int, default=60, help="number of training epochs")
    p.add_argument("--gpu_id", type=str, default="0", help="id(s) for CUDA_VISIBLE_DEVICES")
    p.add_argument("--seed", default=12345, type=int, help="seed for initializing training")
    # optimization
    p.add_argument("--learning_rate", type=float, default=0.05)
    p.add_argument("--lr_decay_epochs", type=str, default="30,40,60")
    p.add_argument("--lr_decay_rate", type=float, default=0.1)
    p.add_argument("--weight_decay", type=float, default=1e-4)
    p.add_argument("--momentum", type=float, default=0.9)
    p.add_argument("--cosine", action="store_true")
    # dataset and model
    p.add_argument("--dataset", type=str, default="prostate_hv")
    p.add_argument("--model_s", type=str, default="effiB0", choices=sorted(model_dict))
    p.add_argument("--model_t", type=str, default="effiB0")
    p.add_argument("--path_t", type=str, default=None, help="teacher model snapshot")
    # augment
    p.add_argument("--aug_train", type=str, default="RA", choices=["NULL", "RA"])
    p.add_argument("--crop", type=float, default=0.2)
    p.add_argument("--image_size", type=int, default=512)
    p.add_argument("--image_resize", action="store_true")
    p.add_argument("--n_cls", type=int, default=8)
    p.add_argument("--skip_test", action="store_true")
    # distillation
    p.add_argument("--trial", type=str, default="1")
    p.add_argument("--kd_T", type=float, default=4)
    p.add_argument("--distill", type=str, default="kd")
    p.add_argument("-c", "--cls", type=float, default=1.0)
    p.add_argument("-d", "--div", type=float, default=1.0)
    p.add_argument("-b", "--beta", type=float, default=0.0)
    p.add_argument("-f", "--factor", type=int, default=2)
    p.add_argument("-s", "--soft", type=float, default=1.0)
    p.add_argument("--hint_layer", default=1, type=int, choices=[0, 1, 2, 3, 4])
    # NCE distillation
    p.add_argument("--feat_dim", default=512, type=int)
    p.add_argument("--mode", default="exact", type=str, choices=["exact", "relax"])
    p.add_argument("--nce_k", default=16384, type=int)
    p.add_argument("--nce_t", default=0.07, type=float)
    p.add_argument("--nce_m", default=0.5, type=float)
    p.add_argument("--alpha", default=0.999, type=float)
    p.add_argument("--mem", default="MoCo", type=str, choices=["MoCo", "MoCoST", "MoCoSSTT", "MoCoAtt"])   # (+MoCoAtt)
    p.add_argument("--head", default="None", type=str, choices=["None", "linear", "mlp"])
    # distill option
    p.add_argument("--weight", type=float, default=1e-4)
    p.add_argument("--std_pre", type=str, default="PANDA")
    p.add_argument("--std_strict", action="store_false", help="strict by default")
    p.add_argument("--tec_pre", type=str, default="ImageNet")
    p.add_argument("--tec_strict", action="store_false", help="strict by default")
    p.add_argument("--attn", type=str, default="self")
    # multiprocessing
    p.add_argument("--dali", type=str, choices=["cpu", "gpu"], default=None)
    p.add_argument("--multiprocessing-distributed", action="store_true")
    p.add_argument("--dist-url", default="tcp://127.0.0.1:23451", type=str)
    p.add_argument("--deterministic", action="store_false")
    p.add_argument("--skip_validation", action="store_false")
    # ---- additions of this implementation ----
    p.add_argument("--moma_prec", default="bf16", choices=["fp32", "bf16"],
                   help="arithmetic of the KD kernels: fp32 = reference arithmetic, bf16 = bf16 MFMA / fp32 accumulate")
    p.add_argument("--queue_dtype", default="fp32", choices=["fp32", "bf16"], help="storage of the K x d queue")
    p.add_argument("--amp", default=None, choices=["bf16", "fp16"], help="autocast dtype for the backbones")
    p.add_argument("--channels_last", action="store_true")
    p.add_argument("--shuffle_bn", default="per_rank", choices=["per_rank", "gather"])
    p.add_argument("--dp", default=None, choices=["flat", "ddp", "auto"],
                   help="student wrap at world size > 1: one flat gradient all-reduce per step (default) or stock DDP")
    p.add_argument("--no_fused", action="store_true", help="reference call sequence on materialised logits")
    p.add_argument("--steps_per_epoch", type=int, default=100, help="synthetic loader length")
    p.add_argument("--num_heads", type=int, default=4)
    p.add_argument("--no_overlap_teacher", dest="overlap_teacher", action="store_false",
                   help="run the teacher / key side of the step on the main stream instead of a second HIP stream")
    p.add_argument("--no_graph_teacher", dest="graph_teacher", action="store_false",
                   help="run the teacher's two no-grad forwards eagerly instead of replaying them from a HIP graph")
    p.add_argument("--no_graph_student", dest="graph_student", action="store_false",
                   help="issue the student forward / backward (and the teacher side, K1, K4) launch by launch instead of replaying "
                        "the step from HIP graphs (helper/step_graph.py)")
    p.add_argument("--reproducible", action="store_true",
                   help="the reference's default semantics (cudnn.deterministic with the seed, no benchmark mode): MIOpen's "
                        "deterministic algorithms, at 3.9x the step time of the default on EfficientNet-B0")
    p.add_argument("--miopen_find", default="on", choices=["on", "off"],
                   help="on = cudnn.benchmark as in the reference (MIOpen find mode: minutes at the first step of a new "
                        "shape); off = immediate mode with the shipped find-db")
    p.add_argument("--save_root", type=str, default="./save")
    p.add_argument("--resume", type=str, default=None,
                   help="checkpoint written by this trainer (ckpt_last.pth): student, EMA teacher, CMO, queue + pointer, optimizer")
    return p


def parse_option(argv=None):
    opt = build_parser().parse_args(argv)
    if opt.distill == "moma":
        opt.nce_t = 0.15                                        # reference :135-136
    opt.moma_fused = not opt.no_fused
    opt.lr_decay_epochs = [int(it) for it in opt.lr_decay_epochs.split(",")]
    # (the reference names the run after torch.cuda.device_count(), train_student_moma.py:147-150.  Counted from sysfs here: this
    #  code also runs in the parent of the ranks, which must not open the device -- see main(); the HIP runtime is asked only
    #  where there is no KFD sysfs to read)
    from .devices import visible_gpu_count
    ngpu = visible_gpu_count()
    if ngpu is None:
        ngpu = torch.cuda.device_count()
    opt.model_path = os.path.join(
        opt.save_root, f"kd_{opt.dataset}_{opt.model_s}_StdPre_{opt.std_pre}_and_TecPre_{opt.tec_pre}"
                       f"_CPU{opt.num_workers}_GPU{ngpu}/")
    opt.tb_path = os.path.join(opt.save_root, "students/")
    name = (f"{opt.distill}_{opt.dataset}_{opt.model_s}_BS{opt.batch_size}_lr_{opt.learning_rate}_decay"
            f"_{opt.weight_decay}_seed{opt.seed}_imageS_{opt.image_size}_cosine_{opt.cosine}"
            f"_StdPre_{opt.std_pre}_strict_{opt.std_strict}_and_TecPre_{opt.tec_pre}_strict_{opt.tec_strict}"
            f"_TB0_SB0_BZ64_attn_{opt.attn}")
    if opt.distill == "moma":
        name = f"{name}_{opt.mem}_head_{opt.head}_{opt.feat_dim}"
    opt.model_name = f"{name}_c{opt.cls}_d{opt.div}_b{opt.beta}_trial_{opt.trial}"
    opt.tb_folder = os.path.join(opt.tb_path, opt.model_name)
    opt.save_folder = os.path.join(opt.model_path, opt.model_name)
    return opt


def build_training(opt, device):
    """Models, criteria, queue, optimizer in the reference's construction (and RNG) order (:263-392).
    Returns (model_s, model_t, module_list, criterion_list, trainable_list, contrast, optimizer)."""
    size = IMAGE_SIZE.get(opt.dataset, opt.image_size)
    data = torch.randn(2, 3, size, size)                         # consumes RNG before model init (:263-268)
    model_s = load_model(opt.model_s, opt.std_pre, opt.n_cls, opt.std_strict, opt.gpu, opt.multiprocessing_distributed)
    model_t = load_model(opt.model_t, opt.path_t or opt.tec_pre, opt.n_cls, opt.tec_strict, opt.gpu,
                         opt.multiprocessing_distributed)
    model_t.eval(); model_s.eval()
    with torch.no_grad():                                        # probe (:274-277) on a small crop to read dims
        probe = data[:, :, :min(size, 64), :min(size, 64)]
        feat_t, _ = model_t(probe, is_feat=True)
        feat_s, _ = model_s(probe, is_feat=True)
    module_list = nn.ModuleList([model_s])
    trainable_list = nn.ModuleList([model_s])
    criterion_cls = nn.CrossEntropyLoss()
    criterion_div = DistillKL(opt.kd_T)
    contrast = None
    if opt.distill == "kd":
        criterion_kd = DistillKL(opt.kd_T)
    elif opt.distill == "moma":
        opt.s_dim = feat_s[-1].shape[1]
        opt.t_dim = feat_t[-1].shape[1]
        if opt.head == "None":
            opt.feat_dim = opt.s_dim
        contrast = build_mem(opt).to(device)                     # randn(K,d) drawn here (:333-334)
        criterion_kd = CMO(opt)
        if opt.head == "mlp":
            module_list.append(criterion_kd.embed_s)
            module_list.append(criterion_kd.embed_t)
            trainable_list.append(criterion_kd.embed_s)
            criterion_kd.embed_t.eval()
        # reference :345-356 lists self_mix / dual / self_nomix / default; the other CMO layouts (all, qk -> `atts`;
        # dual2 -> atts_p/atts_n; self_qk -> atts_q/atts_k) would hit a missing attribute there and are added by name here
        for name in ("atts", "atts_p", "atts_n", "atts_q", "atts_k", "atts_queue"):
            if hasattr(criterion_kd, name):
                trainable_list.append(getattr(criterion_kd, name))
    else:
        raise NotImplementedError(opt.distill)
    criterion_list = nn.ModuleList([criterion_cls, criterion_div, criterion_kd])
    module_list.append(model_t)
    module_list.to(device)
    criterion_list.to(device)
    if getattr(opt, "channels_last", False):
        model_s.to(memory_format=torch.channels_last)
        model_t.to(memory_format=torch.channels_last)
    optimizer = make_optimizer(trainable_list.parameters(), opt, device)      # (after the moves: its state follows the parameters' layout)
    return model_s, model_t, module_list, criterion_list, trainable_list, contrast, optimizer


def make_optimizer(params, opt, device, fused=None):
    """SGD as the reference builds it (train_student_moma.py:389-392).  --amp fp16 on a GPU: torch's FUSED SGD, which takes
    GradScaler's scale and found-inf flag as DEVICE tensors (an overflowed step is skipped inside the kernel), so `scaler.step()`
    reads nothing back -- the stock path syncs the host once per step (`found_inf.item()`), which cost BASELINE configs[4] its
    run-ahead (round 4: 119 ms of host time in a 120 ms step) and kept the step out of HIP graphs.
    The fused kernel's momentum buffers are created HERE, as zeros: torch allocates them uninitialised on the optimizer's first
    step and fills them inside the kernel -- which does nothing when that step overflows, as the first steps under a GradScaler
    do by design (initial scale 2^16): the second step then runs on garbage momentum (seen: configs[4] at full size, NaN weights
    after its second step).  With dampening 0, `buf = momentum * 0 + grad` IS the first-step rule `buf = grad`."""
    params = list(params)
    if fused is None:                              # (an explicit value: the CPU test of this function)
        fused = getattr(opt, "amp", None) == "fp16" and torch.device(device).type == "cuda"
    optimizer = optim.SGD(params, lr=opt.learning_rate, momentum=opt.momentum, weight_decay=opt.weight_decay,
                          **({"fused": True} if fused else {}))
    if fused and opt.momentum != 0:
        for p in params:
            if p.requires_grad:
                optimizer.state[p]["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)
    if fused:
        # torch._fused_sgd_ rewrites the parameters WITHOUT advancing their version counters (checked: `_version` stays put, the
        # plain / foreach paths advance it) -- and everything here that keeps a derived copy of a weight keys it on that counter
        # (the bf16 weight packs of the attention modules, ops.py).  Left alone the eager loop trains atts_q against the packs of
        # its FIRST step (seen: the eager loop learning visibly slower than the graph-served one, which rebuilds the packs inside
        # its graphs).  One host-side call per step, no kernel.
        def _advance_versions(opt_, _args, _kwargs):
            ps = [q for g in opt_.param_groups for q in g["params"] if q.grad is not None]
            if ps:
                torch.autograd.graph.increment_version(ps)
        optimizer.register_step_post_hook(_advance_versions)
    return optimizer


def _rank_state_path(folder, rank):
    return os.path.join(folder, "ckpt_last_rank{}.pth".format(int(rank)))


def _atomic_save(obj, path):
    """torch.save through a temporary name + rename: a job killed mid-write leaves the previous file, never a truncated one."""
    tmp = "{}.tmp{}".format(path, os.getpid())
    torch.save(obj, tmp)
    os.replace(tmp, path)


def _load_rank_state(folder, rank, epoch, device):
    """This rank's queue / pointer / RNG state, or None when the file is absent, unreadable or belongs to ANOTHER epoch than
    the shared checkpoint (a job killed between the two writes, stale files of an earlier run in the same folder): the caller
    then falls back to the shared file's queue and leaves the RNG streams alone."""
    mine = _rank_state_path(folder, rank)
    if not os.path.isfile(mine):
        return None
    try:
        rs = torch.load(mine, map_location=device, weights_only=False)
    except Exception as e:                                  # truncated / foreign file
        print("[moma] per-rank checkpoint {} unreadable ({}: {}); using the shared queue state".format(mine, type(e).__name__, e))
        return None
    if rs.get("epoch") != epoch:
        print("[moma] per-rank checkpoint {} is from epoch {} but the shared checkpoint from epoch {}: ignored "
              "(queue / pointer from the shared file, RNG streams not restored)".format(mine, rs.get("epoch"), epoch))
        return None
    return rs


# The product needs a GPU (no CPU path exists for the kernels).  Host-logic tests that replace the kernel wrappers by
# stand-ins set this module attribute to False; nothing in the product or its CLI does.
_REQUIRE_GPU = True


def _get_rng_state(device):
    return {"python": random.getstate(), "numpy": np.random.get_state(), "torch": torch.get_rng_state(),
            "cuda": torch.cuda.get_rng_state(device) if device.type == "cuda" else None}


def _set_rng_state(st, device):
    random.setstate(st["python"])
    np.random.set_state(st["numpy"])
    torch.set_rng_state(st["torch"].cpu())
    if st.get("cuda") is not None and device.type == "cuda":
        torch.cuda.set_rng_state(st["cuda"].cpu(), device)


def main_worker(gpu, ngpus_per_node, opt):
    opt.gpu = int(gpu)
    opt.gpu_id = int(gpu)
    # node rank: torchrun exports GROUP_RANK (NODE_RANK is the older launcher's name); single node -> 0
    opt.rank = int(os.environ.get("GROUP_RANK", os.environ.get("NODE_RANK", 0)))
    opt.dist_backend = os.environ.get("MOMA_DIST_BACKEND", "nccl")      # 'nccl' = RCCL on ROCm (reference :232)
    if not torch.cuda.is_available() and _REQUIRE_GPU:
        raise RuntimeError("train_student_moma: no GPU visible. The MoMA hot path is a HIP library for gfx950 "
                           "and has no CPU fallback.")
    trainer = ContrastTrainer(opt)
    trainer.init_ddp_environment(gpu, ngpus_per_node)
    if opt.seed is not None:
        random.seed(opt.seed)
        torch.manual_seed(opt.seed)
        np.random.seed(opt.seed)
    # (reference: benchmark on in init_ddp_environment (:34), off again + cudnn.deterministic with a seed (:241-246), on only with
    #  its --deterministic switch (:417-418); here the fast algorithms are the default and --reproducible asks for that mode)
    if getattr(opt, "reproducible", False):
        if opt.seed is None:
            opt.seed = 12345
            random.seed(opt.seed); torch.manual_seed(opt.seed); np.random.seed(opt.seed)
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
        print("reproducible: MIOpen deterministic algorithms, seed {} (bitwise repeatable inside a process; MIOpen's solver choice "
              "may differ between processes)".format(opt.seed))
    else:
        torch.backends.cudnn.benchmark = opt.miopen_find == "on"
    device = torch.device("cuda", opt.gpu) if torch.cuda.is_available() else torch.device("cpu")
    opt.device = device
    print("opt.n_cls: ", opt.n_cls)

    model_s, model_t, module_list, criterion_list, trainable_list, contrast, optimizer = build_training(opt, device)
    if contrast is not None:
        trainer.broadcast_memory(contrast)                       # optional step: synchronize memory (:336)
    if opt.multiprocessing_distributed:
        from .learning.ddp import wrap_student
        ddp_s = wrap_student(model_s, device_ids=[opt.gpu] if device.type == "cuda" else None, mode=getattr(opt, "dp", None))
        module_list = [ddp_s] + [m for m in list(module_list)[1:]]
        # criterion modules and the teacher are under no wrap: the reference relies on identical seeds; one flat
        # broadcast from rank 0 makes it a fact
        from .learning.ddp import broadcast_module_state
        broadcast_module_state([criterion_list[2], model_t])
    if opt.amp == "fp16":
        opt._grad_scaler = torch.amp.GradScaler("cuda")
    print("opt.batch_size", opt.batch_size)

    size = IMAGE_SIZE.get(opt.dataset, opt.image_size)
    seed = (opt.seed or 0) + opt.rank
    train_loader = SyntheticLoader(opt.steps_per_epoch, opt.batch_size, size, opt.n_cls, seed, device)
    val_loader = SyntheticLoader(max(1, opt.steps_per_epoch // 10), opt.batch_size, size, opt.n_cls, seed + 1, device)
    is_main = (not opt.multiprocessing_distributed) or opt.rank % ngpus_per_node == 0
    if is_main:
        os.makedirs(opt.save_folder, exist_ok=True)
        trainer.args.tb_folder = opt.tb_folder
    best_acc, best_f1, t_total = 0.0, 0.0, time.time()
    start_epoch = 1
    if opt.resume:
        ck = torch.load(opt.resume, map_location=device, weights_only=False)
        model_s.load_state_dict(ck["model"])
        model_t.load_state_dict(ck["model_t"])
        criterion_list[2].load_state_dict(ck["criterion_kd"])
        optimizer.load_state_dict(ck["optimizer"])
        if getattr(opt, "_grad_scaler", None) is not None and ck.get("grad_scaler"):
            opt._grad_scaler.load_state_dict(ck["grad_scaler"])         # (--amp fp16: the loss scale continues where it was)
        # per-rank state (the queue and its pointer are per rank in the default per_rank mode, and so are the RNG streams)
        # sits next to the shared file; a run resumed on fewer / other ranks falls back to rank 0's queue
        rs = _load_rank_state(os.path.dirname(opt.resume), opt.rank, ck["epoch"], device)
        if contrast is not None:
            qstate = (rs or {}).get("contrast") or ck.get("contrast")
            if qstate is not None:
                contrast.load_state_dict(qstate)
        if rs is not None and rs.get("rng") is not None:
            _set_rng_state(rs["rng"], device)
        best_acc, best_f1, start_epoch = ck.get("best_acc", 0.0), ck.get("best_f1", 0.0), ck["epoch"] + 1
        print("==> resumed from {} (epoch {}, queue pointer {}, per-rank state {})".format(
            opt.resume, ck["epoch"], contrast.index if contrast is not None else "-", "found" if rs else "absent"))
    for epoch in range(start_epoch, opt.epochs + 1):
        adjust_learning_rate(epoch, opt, optimizer)
        print("==> training...")
        t1 = time.time()
        train_acc, train_loss = train_distill_moma(epoch, train_loader, module_list, criterion_list,
                                                   trainer if opt.distill == "moma" else None, contrast, optimizer, opt)
        if device.type == "cuda":
            torch.cuda.synchronize()
        t2 = time.time()
        if opt.multiprocessing_distributed:
            metrics = torch.tensor([train_acc, train_loss], device=device)
            train_acc, train_loss = reduce_tensor(metrics, opt.world_size).tolist()
        if is_main:
            ips = opt.steps_per_epoch * opt.batch_size * max(1, getattr(opt, "world_size", 1)) / (t2 - t1)
            print(" * Epoch {}, Acc@1 {:.3f}, Loss {:.4f}, Time {:.2f}, {:.1f} images/sec".format(
                epoch, train_acc, train_loss, t2 - t1, ips))
        val_acc, val_loss, val_stat = validate_distill(val_loader, module_list, criterion_list[0], opt, prefix="Val")
        test_acc = test_loss = test_stat = None
        if not opt.skip_test:                                    # reference :515-519 (the synthetic set doubles as test)
            test_acc, test_loss, test_stat = validate_distill(val_loader, module_list, criterion_list[0], opt, prefix="Test")
        if is_main:
            val_f1 = macro_f1(val_stat["conf_mat"])               # model selection by accuracy AND by macro-F1 (:522-571)
            print(" ** Acc_val@1 {:.3f}  F1_val {:.4f}".format(val_acc, val_f1))
            if test_acc is not None:
                print(" ** Acc_test@1 {:.3f}".format(test_acc))
            state = {"epoch": epoch, "model": model_s.state_dict(), "best_acc": best_acc,
                     "optimizer": optimizer.state_dict()}
            if val_acc > best_acc:
                best_acc = val_acc
                state.update(best_acc=best_acc, best_acc_epoch=epoch)
                print("saving the best acc model!")
                _atomic_save(state, os.path.join(opt.save_folder, "net_best_acc.pth"))
            if val_f1 > best_f1:
                best_f1 = val_f1
                state.update(best_f1=best_f1, best_f1_epoch=epoch)
                print("saving the best f1 model!")
                _atomic_save(state, os.path.join(opt.save_folder, "net_best_f1.pth"))
            metrics = {"val_cf": val_stat["conf_mat"].tolist(), "val_loss": val_loss, "val_acc": val_acc}
            if test_stat is not None:
                metrics.update(test_cf=test_stat["conf_mat"].tolist(), test_loss=test_loss, test_acc=test_acc)
            update_dict_to_json(epoch, metrics, os.path.join(opt.save_folder, "stat.json"))
            # full training state (the reference saves the student only): shared part by the main rank ...
            _atomic_save({"epoch": epoch, "model": model_s.state_dict(), "model_t": model_t.state_dict(),
                          "criterion_kd": criterion_list[2].state_dict(),
                          "contrast": contrast.state_dict() if contrast is not None else None,
                          "optimizer": optimizer.state_dict(), "best_acc": best_acc, "best_f1": best_f1,
                          "grad_scaler": opt._grad_scaler.state_dict() if getattr(opt, "_grad_scaler", None) is not None else None},
                         os.path.join(opt.save_folder, "ckpt_last.pth"))
        # ... and the per-rank part (queue + pointer of THIS rank, RNG streams) by every rank; both carry the epoch and a
        # resume only pairs files of the same epoch (_load_rank_state)
        os.makedirs(opt.save_folder, exist_ok=True)
        _atomic_save({"epoch": epoch, "contrast": contrast.state_dict() if contrast is not None else None,
                      "rng": _get_rng_state(device)}, _rank_state_path(opt.save_folder, opt.rank))
        if opt.multiprocessing_distributed and torch.distributed.is_initialized():
            torch.distributed.barrier()                         # no rank starts the next epoch before every file of this one exists
    if is_main:
        print("best accuracy:", best_acc)
        save_state = {k: v for k, v in vars(opt).items() if not k.startswith("_") and k != "trace"}
        save_state["Total params"] = sum(p.numel() for p in model_s.parameters()) / 1e6
        save_state["Total time"] = (time.time() - t_total) / 3600.0
        save_dict_to_json(save_state, os.path.join(opt.save_folder, "parameters.json"))
    if opt.multiprocessing_distributed and torch.distributed.is_initialized():
        torch.distributed.barrier()                             # (rank 0 has written its files before any rank tears the group down)
        torch.distributed.destroy_process_group()


def parent_gpu_count(gpu_id: str) -> int:
    """Ranks to spawn, decided without a HIP call: the GPUs sysfs shows (moma_amd.devices.visible_gpu_count); where there is no KFD
    sysfs to read, the length of the --gpu_id list the user gave (a rank whose device is missing says so itself)."""
    from .devices import visible_gpu_count
    n = visible_gpu_count()
    if n is None:
        n = len([x for x in gpu_id.split(",") if x.strip() != ""])
    if n < 1:
        raise SystemExit("train_student_moma: --multiprocessing-distributed found no GPU to start a rank on")
    return n


def main(argv=None):
    opt = parse_option(argv)
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:         # launched by torchrun: 1 process / GPU
        opt.multiprocessing_distributed = True
        opt.world_size = int(os.environ["WORLD_SIZE"])
        n_local = int(os.environ.get("LOCAL_WORLD_SIZE", opt.world_size))
        main_worker(int(os.environ.get("LOCAL_RANK", 0)), n_local, opt)
        return
    os.environ["CUDA_VISIBLE_DEVICES"] = opt.gpu_id
    if opt.multiprocessing_distributed:
        # The reference's launch mode (train_student_moma.py:207-224: `--distill moma` needs it).  This process only starts the
        # ranks: it counts the devices from KFD's sysfs topology (narrowed by the CUDA_VISIBLE_DEVICES just set) and never asks
        # the HIP runtime -- torch.cuda.device_count() ends in hipGetDeviceCount where amdsmi is unusable, and a parent that has
        # opened the device is one more process on the card and may not start other programs on this platform.
        import torch.multiprocessing as mp
        ngpus_per_node = parent_gpu_count(opt.gpu_id)
        opt.ngpus_per_node = ngpus_per_node
        opt.world_size = ngpus_per_node                            # single node, as the reference (:218-219)
        mp.spawn(main_worker, nprocs=ngpus_per_node, args=(ngpus_per_node, opt))
    else:
        ngpus_per_node = torch.cuda.device_count()                 # (this process IS the worker)
        opt.ngpus_per_node = ngpus_per_node
        opt.world_size = 1
        main_worker(0, ngpus_per_node, opt)


if __name__ == "__main__":
    main()
