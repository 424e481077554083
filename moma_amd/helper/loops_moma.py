"""The MoMA contrastive-distillation step loop (reference: helper/loops_moma.py:221-373, moma branch
:308-335) and a minimal validation loop (:448-529).

Order of operations per batch is the reference's: student fwd, teacher fwd #1, CE + KL terms,
EMA of the teacher (K4), Shuffle-BN key encoding (teacher fwd #2), heads, batch-token attention (K1),
InfoNCE over the queue + enqueue (K2, K3), weighted sum, backward, optimizer step.  Differences, all
performance-only: the per-step `.item()` host syncs (:351,355) are deferred to print time; `atts_k` /
`atts_queue` run under no_grad (their outputs are detached / enqueued under no_grad in the reference, so
their grads are None either way -- Q6); the KD term uses the one-pass fused kernel unless
opt.moma_fused is False, in which case the reference call sequence contrast(...) -> CrossEntropyLoss runs
on materialised logits.
"""
from __future__ import print_function

import sys
import time

import numpy as np
import torch
import torch.nn as nn

from .util import AverageMeter, accuracy
from ..learning.contrast_trainer import ContrastTrainer
from ..learning.ddp import FlatDataParallel


def _set_bn_train(m):
    if m.__class__.__name__.find("BatchNorm") != -1:
        m.train()


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def _same_arch(a, b):
    pa, pb = list(a.parameters()), list(b.parameters())
    return len(pa) == len(pb) and all(x.shape == y.shape for x, y in zip(pa, pb))


class MomaStep:
    """One training step of the moma / kd branch, cut where HIP graphs can be cut (helper/step_graph.py):

        student_forward   the student's forward                                                    -- capturable, main stream
        teacher_side      teacher forward #1, EMA (K4), Shuffle-BN key encoding, key-side attention (K1)
                                                                                                   -- capturable, side stream
        losses_and_query  CE + KL, embed_s, atts_q (K1; leaves q packed for K2), the student's top-1 -- capturable, after the join
        kd_term           K2 + K3 (+ the host pointer): contrast.forward_fused (autograd) or, between graph replays,
                          contrast.forward_fused_into (static buffers)                              -- always eager
        backward_part     weighted sum, backward                                                    -- capturable

    `run_eager` is these in sequence -- the loop body of the reference (helper/loops_moma.py:256-361) in its order."""

    def __init__(self, module_list, criterion_list, trainer, contrast, optimizer, opt, dev):
        self.criterion_cls, self.criterion_div, self.criterion_kd = criterion_list[0], criterion_list[1], criterion_list[2]
        self.model_s, self.model_t = module_list[0], module_list[-1]
        self.trainer, self.contrast, self.optimizer, self.opt, self.dev = trainer, contrast, optimizer, opt, dev
        self.amp_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(getattr(opt, "amp", None))
        self.scaler = getattr(opt, "_grad_scaler", None)
        self.fused = getattr(opt, "moma_fused", True)
        self.mocoatt = getattr(opt, "mem", "MoCo") == "MoCoAtt"
        self.attn_in_shuffle = (opt.distill == "moma" and getattr(opt, "attn", "self") in ("self_mix", "self_nomix")
                                and not self.mocoatt)
        self.ema_ok = None
        if opt.distill == "moma" and contrast is not None and hasattr(contrast, "memory_s"):
            # --mem MoCoST / MoCoSSTT: the reference loop calls contrast(q=f_s, k=k, all_k=all_k) for every memory
            # (helper/loops_moma.py:331) and never produces the second key set k_t these two need -- there it dies inside the first
            # step (MoCoST.forward: missing argument k_t; MoCoSSTT.forward: None.detach()).  Same verdict here, said up front:
            raise NotImplementedError(f"--mem {type(contrast).__name__}: the distillation loop supplies (q, k, all_k) only -- the "
                                      "reference's own loop fails on this memory in its first step (no k_t is ever computed); "
                                      "the dual-queue memories are usable as modules (forward / forward_fused), not from this loop")
        self.overlap = (getattr(opt, "overlap_teacher", False) and opt.distill == "moma" and dev.type == "cuda"
                        and getattr(opt, "shuffle_bn", "per_rank") == "per_rank")

    def autocast(self):
        return torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype is not None)

    def graphable(self):
        """What helper/step_graph.py captures: the bench / run-script configuration (--distill moma, one-pass K2, --attn self /
        self_mix / self_nomix, MoCo memory, per-rank Shuffle-BN; fp16 + GradScaler when the optimizer takes the scale and the found-inf flag on the
        device -- torch's fused SGD, what build_training makes for --amp fp16: the captured backward multiplies by the scaler's
        device tensor, `scaler.step / update` stay eager behind the graphs and read nothing back).  Everything else keeps the
        eager loop."""
        o = self.opt
        scaler_ok = self.scaler is None or bool(getattr(self.optimizer, "_step_supports_amp_scaling", False))
        # (round 6: also --attn self_mix / self_nomix -- the key encoding with the attention in front of the un-shuffle,
        #  learning/contrast_trainer.py:_shuffle_bn_attn, runs inside g_query behind the student's forward; K2 packs q itself)
        if not (self.dev.type == "cuda" and o.distill == "moma" and scaler_ok and getattr(o, "shuffle_bn", "per_rank") == "per_rank"):
            return False
        if self.mocoatt:
            # --mem MoCoAtt (round 6): the memory's cross-attention variant, the materialised logits and their CrossEntropy are
            # captured with the query side (MoCoAtt.forward_logits); only the enqueue stays between the graphs
            return getattr(o, "attn", "self") != "dual2" and hasattr(self.contrast, "forward_logits")
        return (self.fused and getattr(o, "attn", "self") in ("self", "self_mix", "self_nomix")
                and hasattr(self.contrast, "forward_fused_into"))

    def teacher_side(self, images, teacher):
        """teacher forward #1 (:270-272), then the moma branch's no-grad part (:309-320, :327-329)."""
        opt, trainer, criterion_kd, model_t = self.opt, self.trainer, self.criterion_kd, self.model_t
        with self.autocast(), torch.no_grad():
            _, lt = teacher(images, is_feat=True)
        if opt.distill != "moma":
            return lt.float(), None, None
        student = _unwrap(self.model_s)
        if self.ema_ok is None:
            self.ema_ok = _same_arch(student, model_t)
            if not self.ema_ok and getattr(opt, "rank", 0) == 0:
                # the reference raises half-way through the zip; defined behaviour here:
                print("[moma] student/teacher architectures differ: teacher stays frozen (no EMA)")
        if self.ema_ok:
            trainer.momentum_update(student, model_t, opt.alpha)                     # K4 (:309)
            if opt.head == "mlp":
                criterion_kd.embed_t.eval()
                if _same_arch(criterion_kd.embed_s, criterion_kd.embed_t):
                    trainer.momentum_update(criterion_kd.embed_s, criterion_kd.embed_t, opt.alpha)
        model_t.apply(_set_bn_train)                                                  # (:314-318)
        if self.attn_in_shuffle:         # key encoding + attention need the student's query: done on the main stream below
            return lt.float(), None, None
        with self.autocast():
            kk, akk = trainer._shuffle_bn(images, teacher, model_ema_head=criterion_kd.embed_t)   # (:320)
        kk, akk = kk.float(), akk.float()
        if opt.attn == "self" and not self.mocoatt:                                   # K1, key side (:327-329)
            with torch.no_grad():           # one group of launches for the two key-side modules
                kk, akk = criterion_kd.atts_k.forward_group([criterion_kd.atts_k, criterion_kd.atts_queue], [kk, akk])
        return lt.float(), kk, akk

    def side_stream(self):
        """the second HIP stream of the step (opt.overlap_teacher), None when the teacher side runs on the main stream"""
        if not self.overlap:
            return None
        side = getattr(self.trainer, "_side_stream", None)
        if side is None:
            side = self.trainer._side_stream = torch.cuda.Stream(device=self.dev)
        return side

    def student_forward(self, images, student=None):
        """(:268) -> (feat_s, logit_s)"""
        student = self.model_s if student is None else student
        with self.autocast():
            return student(images, is_feat=True)

    def losses_and_query(self, feat_s, logit_s, logit_t, k, all_k, images, labels, teacher, qpack=None, prefetch=True):
        """CE + KL (:278-279), embed_s (:323-324), the query side of K1 (:326), the student's top-1
        -> dict(loss_cls, loss_div, f_s, k, all_k, acc, qp)"""
        opt, trainer, criterion_kd, contrast = self.opt, self.trainer, self.criterion_kd, self.contrast
        logit_s = logit_s.float()
        out = {"loss_cls": self.criterion_cls(logit_s, labels), "loss_div": self.criterion_div(logit_s, logit_t), "qp": None}
        f_s = None
        if opt.distill == "moma":
            with self.autocast():
                f_s = criterion_kd.embed_s(feat_s[-1])                                    # (:323-324)
            f_s = f_s.float()
            if self.attn_in_shuffle:
                # attn in {self_mix, self_nomix}: Shuffle-BN key encoding with the attention applied before the un-shuffle,
                # over [q ; k] or per side (reference learning/contrast_trainer.py:135-187; train_student_moma.py:345-352
                # registers these modules as trainable -- the reference loop never reaches the call)
                with self.autocast():
                    f_s, k, all_k = trainer._shuffle_bn_attn(images, teacher, criterion_kd.embed_t, criterion_kd, f_s)
                f_s, k, all_k = f_s.float(), k.float(), all_k.float()
            elif opt.attn == "self" and not self.mocoatt:                                 # K1, query side (:326)
                if prefetch:
                    self.prefetch_queue()
                # atts_q's proj epilogue also leaves q in the packed bf16 layout K2 loads it in (no pre-pack launch in K2)
                if qpack is None and self.fused and hasattr(contrast, "qpack"):
                    qpack = contrast.qpack(f_s.shape[0], f_s.shape[1], f_s.device)
                f_s = criterion_kd.atts_q(f_s, qpack=qpack)
                out["qp"] = qpack
        out.update(f_s=f_s, k=k, all_k=all_k, acc=accuracy(logit_s, labels, topk=(1,))[0].squeeze(0))
        return out

    def prefetch_queue(self):
        """opt.prefetch_queue (default OFF since round 5): the queue was last read a whole step ago: sweep it into the Infinity
        Cache on the side stream while the (launch-latency-bound) attention module runs on the main one.  Measured (round 4): the
        sweep re-reads the whole queue (67 MB, 15 us on the side stream) to take 0.6 us off the one-pass kernel -- 2.5x the
        algorithmic bytes of the KD term for nothing the step's time shows; kept as an experiment switch only."""
        if self.overlap and self.fused and getattr(self.opt, "prefetch_queue", False) and hasattr(self.contrast, "prefetch"):
            self.contrast.prefetch(stream=self.side_stream())

    def forward_part(self, images, labels, teacher):
        """-> dict(loss_cls, loss_div, f_s, k, all_k, acc, qp).  opt.overlap_teacher: everything on the teacher / key side of the
        step (all no-grad) is queued on a second HIP stream and runs concurrently with the student forward; the streams join
        before the losses.  Same operations in the same order per stream as the sequential loop -- the two chains do not
        depend on each other until the loss."""
        main_stream = torch.cuda.current_stream() if self.dev.type == "cuda" else None
        side = self.side_stream()
        if side is not None:
            side.wait_stream(main_stream)                     # last step's optimizer, this step's images
            with torch.cuda.stream(side):
                logit_t, k, all_k = self.teacher_side(images, teacher)
        feat_s, logit_s = self.student_forward(images)
        if side is not None:
            main_stream.wait_stream(side)
            for t in (logit_t, k, all_k):
                if t is not None:
                    t.record_stream(main_stream)
        else:
            logit_t, k, all_k = self.teacher_side(images, teacher)
        return self.losses_and_query(feat_s, logit_s, logit_t, k, all_k, images, labels, teacher)

    def kd_term(self, fw):
        """K2 + K3 through autograd (the eager step): -> loss_kd"""
        opt, trainer, contrast, criterion_kd = self.opt, self.trainer, self.contrast, self.criterion_kd
        if opt.distill == "kd":
            return 0
        if opt.distill != "moma":
            raise NotImplementedError(opt.distill)
        f_s, k, all_k, qp = fw["f_s"], fw["k"], fw["all_k"], fw["qp"]
        if self.mocoatt:
            # --mem MoCoAtt: the memory applies the teacher-student cross-attention variant itself
            # (reference MoMA/mem_moco.py:111-161); materialised logits -> CrossEntropy as in :331-335
            if opt.attn == "dual2":
                raise NotImplementedError("attn='dual2' yields positive logits only ([B]); the reference's CrossEntropy "
                                          "over them is undefined (MoMA/mem_moco.py:51-66,148-149)")
            criterion = nn.CrossEntropyLoss()
            output = contrast(q=f_s, k=k, all_k=all_k, attn=opt.attn, criterion_kd=criterion_kd)
            c_losses, _ = trainer._compute_loss_accuracy(logits=output[:-1], target=output[-1], criterion=criterion)
            return c_losses[0]
        if self.fused:                                                                    # K2 + K3
            loss_kd, _acc_kd = contrast.forward_fused(f_s, k, all_k, **({"qpack": qp} if qp is not None else {}))
            return loss_kd
        criterion = nn.CrossEntropyLoss()                                                 # reference sequence (:331-335)
        output = contrast(q=f_s, k=k, all_k=all_k)
        c_losses, _ = trainer._compute_loss_accuracy(logits=output[:-1], target=output[-1], criterion=criterion)
        return c_losses[0]

    def kd_logits_loss(self, fw):
        """--mem MoCoAtt, the part of kd_term in front of the enqueue: cross-attention variant + logits from a snapshot of the
        queue + CrossEntropy (reference MoMA/mem_moco.py:111-147, helper/loops_moma.py:331-335) -> loss_kd.  The graph-served step
        captures this and issues `contrast.enqueue_keys` itself."""
        opt, contrast = self.opt, self.contrast
        logits, labels0, _k = contrast.forward_logits(q=fw["f_s"], k=fw["k"], attn=opt.attn, criterion_kd=self.criterion_kd)
        c_losses, _ = self.trainer._compute_loss_accuracy(logits=[logits], target=labels0, criterion=nn.CrossEntropyLoss())
        return c_losses[0]

    def backward_part(self, fw, loss_kd):
        """weighted sum (:350) and backward (:359-360); gradients start from None (zero_grad(set_to_none=True), the reference's call)"""
        opt = self.opt
        loss = opt.cls * fw["loss_cls"] + opt.div * fw["loss_div"] + opt.beta * loss_kd
        self.optimizer.zero_grad(set_to_none=True)
        if self.scaler is not None:
            self.scaler.scale(loss).backward()
        else:
            loss.backward()
        return loss

    def run_eager(self, images, labels, teacher):
        fw = self.forward_part(images, labels, teacher)
        loss_kd = self.kd_term(fw)
        loss = self.backward_part(fw, loss_kd)
        return loss.detach(), (loss_kd.detach() if torch.is_tensor(loss_kd) else loss_kd), fw["acc"]


def train_distill_moma(epoch, train_loader, module_list, criterion_list, trainer, contrast, optimizer, opt):
    """one epoch distillation; returns (top1.avg, losses.avg) like the reference."""
    for module in module_list:
        module.train()
    module_list[-1].eval()                      # teacher in eval for its first forward (:227)

    criterion_kd = criterion_list[2]
    model_s, model_t = module_list[0], module_list[-1]

    batch_time, losses, top1 = AverageMeter(), AverageMeter(), AverageMeter()
    n_batch = len(train_loader)
    scaler = getattr(opt, "_grad_scaler", None)
    dev = getattr(opt, "device", None)
    if dev is None:
        dev = torch.device("cuda", opt.gpu if (opt.gpu is not None and opt.multiprocessing_distributed) else 0) \
            if torch.cuda.is_available() else torch.device("cpu")
    step = MomaStep(module_list, criterion_list, trainer, contrast, optimizer, opt, dev)
    single_rank = bool(getattr(trainer, "grad_sync_single_rank", False))
    sync_criterion = opt.distill == "moma" and (getattr(opt, "world_size", 1) > 1 or single_rank)
    flat_dp = isinstance(model_s, FlatDataParallel)
    flat_params = None
    if flat_dp:
        # learning/ddp.py: ONE flat all-reduce per step behind the backward -- the student's gradients and those of the trainable
        # criterion modules (atts_q / embed_s: not under any wrapper, un-synchronised in the reference)
        flat_params = model_s.grad_params()
        if opt.distill == "moma":
            flat_params = flat_params + [p for p in criterion_kd.parameters() if p.requires_grad]
    elif sync_criterion:
        # stock DDP on the student: one flat all-reduce per step for the trainable criterion modules, launched from autograd
        # hooks (overlaps the backward)
        trainer.attach_grad_sync([p for p in criterion_kd.parameters() if p.requires_grad])
    # the teacher's two no-grad forwards per step are replayed from a HIP graph after a few eager calls
    # (opt.graph_teacher, default on for GPU runs; see helper/graphs.py); everything else uses `model_t` itself
    teacher = model_t
    holder = trainer if trainer is not None else opt          # (--distill kd runs without a ContrastTrainer)
    if getattr(opt, "graph_teacher", True) and dev.type == "cuda":
        teacher = getattr(holder, "_graphed_teacher", None)
        if teacher is None or teacher.module is not model_t:
            from .graphs import GraphedInference
            teacher = holder._graphed_teacher = GraphedInference(model_t)
    # the step itself -- student forward + backward, the teacher side, K1, K4 -- replayed from two HIP graphs with K2 / K3 called
    # between them (opt.graph_student, default on where MomaStep.graphable(); helper/step_graph.py).  Stock DDP's reducer lives
    # in autograd hooks and cannot be captured: the flat wrap (one collective per step, outside the graphs) or a single rank.
    runner = None
    if getattr(opt, "graph_student", True) and step.graphable() and (flat_dp or not hasattr(model_s, "module")):
        from .step_graph import StepGraphs
        runner = getattr(holder, "_step_graphs", None)
        if runner is None or not runner.same_objects(step):
            runner = holder._step_graphs = StepGraphs()
        runner.bind(step)
    elif (getattr(opt, "graph_student", True) and step.graphable() and epoch <= 1 and getattr(opt, "rank", 0) == 0
          and getattr(opt, "graph_student_notice", True)):
        print("[moma] --dp ddp: the stock reducer works from autograd hooks, the student step stays eager (no HIP graph)")
    if (epoch <= 1 and getattr(opt, "overlap_teacher", False) and not step.overlap and dev.type == "cuda"
            and opt.distill == "moma" and getattr(opt, "rank", 0) == 0):
        print("[moma] --shuffle_bn gather: the teacher side stays on the main stream (its collectives are ordered "
              "with the DDP all-reduce there); overlap_teacher is off in this mode")
    trace = getattr(opt, "trace", None)
    step_events = getattr(opt, "step_events", None)     # optional (bench.py): (host time, HIP event) at the end of every step

    end = time.time()
    for idx, data in enumerate(train_loader):
        images, labels = data
        images = images.to(dev, non_blocking=True)
        labels = labels.to(dev, non_blocking=True)
        if getattr(opt, "channels_last", False):
            images = images.contiguous(memory_format=torch.channels_last)

        # =================== forward, KD term, backward =====================
        res = runner.step(images, labels) if runner is not None else None      # None: this step is not (yet) served from graphs
        if res is None:
            res = step.run_eager(images, labels, teacher)
        loss, loss_kd, acc = res
        losses.update(loss, images.size(0))
        if trace is not None:           # optional per-step record (tests / benchmarking), device tensors
            trace.append((loss, contrast.index if contrast is not None else None, loss_kd))

        # =================== metrics =====================
        top1.update(acc, images.size(0))

        # =================== gradient exchange, optimizer =====================
        if flat_dp:
            n_red = model_s.allreduce_grads(flat_params, single_rank=single_rank)
            if trainer is not None and n_red:
                trainer.grad_sync_launches = getattr(trainer, "grad_sync_launches", 0) + n_red
        elif sync_criterion:
            trainer.finish_grad_sync()                  # atts_q / embed_s are not under DDP (fixes Q7)
        if scaler is not None:
            scaler.step(optimizer)
            scaler.update()
        else:
            optimizer.step()

        batch_time.update(time.time() - end)
        end = time.time()
        if step_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_events.append((time.perf_counter(), ev, time.thread_time()))

        if idx % opt.print_freq == 0:
            print("Epoch: [{0}][{1}/{2}]\tGPU {3}\tTime: {bt:.3f}\tLoss {loss:.4f}\tAcc@1 {acc:.3f}".format(
                epoch, idx, n_batch, opt.gpu, bt=batch_time.avg, loss=float(losses.avg), acc=float(top1.avg)))
            sys.stdout.flush()
    return float(top1.avg), float(losses.avg)


def macro_f1(conf_mat):
    """Mean per-class F1 from a confusion matrix (rows = true, columns = predicted); a class with no true positive
    counts 0 (reference train_student_moma.py:522-531)."""
    cm = np.asarray(conf_mat, dtype=np.float64)
    f = 0.0
    for i in range(cm.shape[0]):
        if cm[i, i] > 0:
            prec, rec = float(cm[i, i] / cm[:, i].sum()), float(cm[i, i] / cm[i, :].sum())
            f += 2 * prec * rec / (prec + rec)
    return float(f / cm.shape[0])


def validate_distill(val_loader, module_list, criterion, opt, prefix="Test"):
    """Evaluation of the student (reference :448-529): returns (top-1 %, mean loss, {'acc', 'conf_mat'}).
    The confusion matrix is accumulated on the GPU (one bincount per batch) instead of gathering all logits on the
    host; under DDP the sums, counts and the matrix are summed over the ranks as in the reference (:512-527)."""
    model = module_list[0] if isinstance(module_list, (list, tuple, torch.nn.ModuleList)) else module_list
    for m in (module_list if isinstance(module_list, (list, tuple, torch.nn.ModuleList)) else [module_list]):
        m.eval()
    n_cls = int(opt.n_cls)
    dev = next(model.parameters()).device
    conf = torch.zeros(n_cls * n_cls, dtype=torch.int64, device=dev)
    sums = torch.zeros(4, dtype=torch.float64, device=dev)          # top-1 sum, loss sum, count, count
    with torch.no_grad():
        for idx, (images, labels) in enumerate(val_loader):
            images = images.to(dev, non_blocking=True)
            labels = labels.to(dev, non_blocking=True)
            output = model(images).float()
            n = images.size(0)
            pred = output.argmax(dim=1)
            sums[0] += (pred == labels).sum() * 100.0
            sums[1] += criterion(output, labels).detach().double() * n
            sums[2] += n
            conf += torch.bincount(labels * n_cls + pred, minlength=n_cls * n_cls)
            if idx % opt.print_freq == 0 and idx > 0:
                s = sums.tolist()
                print("{}: [{}/{}]\tGPU: {}\tLoss {:.4f}\tAcc@1 {:.3f}".format(prefix, idx, len(val_loader), opt.gpu,
                                                                              s[1] / s[2], s[0] / s[2]))
    if getattr(opt, "multiprocessing_distributed", False) and torch.distributed.is_initialized():
        torch.distributed.all_reduce(sums)
        torch.distributed.all_reduce(conf)
    s = sums.tolist()
    acc, loss = s[0] / max(s[2], 1.0), s[1] / max(s[2], 1.0)
    return acc, loss, {"acc": acc, "conf_mat": conf.view(n_cls, n_cls).cpu().numpy()}
