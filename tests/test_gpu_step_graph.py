"""HIP-graph replay of the training step (moma_amd/helper/step_graph.py) and of the teacher forwards (helper/graphs.py): the graphs
change WHO issues the launches, never what is computed.  The loop tests of tests/test_gpu_step.py run with the step graphs on
(the default) against the reference's traces; here the graph path is compared with the eager loop directly, its host-side
bookkeeping is checked (queue pointer, BatchNorm batch counters, gradient re-attachment across interleaved eager steps, the
Shuffle-BN permutation stream), and the capture hazards found in round 3 get their regression tests."""
import argparse

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _run(*args, **kw):
    """The loop under MIOpen's deterministic (default) algorithms: its searched weight-gradient kernels are not bitwise reproducible,
    and since round 6 nothing in this library is not (the last fp32 atomics, in the gradient of the materialised-logits paths, went
    then) -- so two runs of the same loop are bit-identical (scripts/diag_graph_noise_tail.py: every variant below, eager against
    eager and graph-served against eager, 0.0 in every loss, weight and queue row), and the comparisons below are exact."""
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        return _run_impl(*args, **kw)
    finally:
        torch.backends.cudnn.deterministic = det


class _SyncTail:
    def __init__(self, items, n, epoch):
        self.items, self.n, self.epoch = items, n, epoch

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        for i, it in enumerate(self.items):
            if i >= len(self.items) - self.n:
                torch.cuda.synchronize()
                print(f"    epoch {self.epoch}: everything before step {i} of {len(self.items)} has finished on the device", flush=True)
            yield it


def _same(a, b):
    """graph-served == eager, bit for bit: every per-step loss, the queue, student, EMA teacher, attention weights, the update"""
    assert np.array_equal(a["loss"], b["loss"]), np.abs(a["loss"] - b["loss"]).max()
    assert np.array_equal(a["loss_kd"], b["loss_kd"]), np.abs(a["loss_kd"] - b["loss_kd"]).max()
    assert np.array_equal(a["memory"], b["memory"]) and np.array_equal(a["atts_q"], b["atts_q"])
    assert np.linalg.norm(b["delta"]) > 0 and np.array_equal(a["delta"], b["delta"])
    for name in a["student"]:
        assert np.array_equal(a["student"][name], b["student"][name]) and np.array_equal(a["teacher"][name], b["teacher"][name]), name


def _run_impl(graph_student, model, overlap, prec, queue_dtype, amp, epochs=2, steps=7, B=8, K=256, d=64, size=32, lr=0.02, scale0=2.0 ** 16,
              attn="self", mem="MoCo", data_on_device=False, validate=False, print_freq=1000, sync_tail=0, graph_teacher=True):
    from moma_amd.backbones import model_dict
    from moma_amd.MoMA.mem_moco import build_mem
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.distiller_zoo import DistillKL

    torch.backends.cudnn.benchmark = False
    dev = torch.device("cuda", 0)
    torch.manual_seed(11)
    kw = {"dropout_rate": 0.0, "drop_connect_rate": 0.0} if model == "effiB0" else {}
    ms, mt = model_dict[model](num_classes=10, **kw), model_dict[model](num_classes=10, **kw)
    with torch.no_grad():
        s_dim = ms.eval()(torch.randn(2, 3, size, size), is_feat=True)[0][-1].shape[1]
    opt = argparse.Namespace(distill="moma", head="mlp", feat_dim=d, attn=attn, mem=mem, nce_k=K, nce_t=0.15, alpha=0.99,
                             cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=0, multiprocessing_distributed=False, print_freq=print_freq,
                             batch_size=B, rank=0, world_size=1, s_dim=s_dim, t_dim=s_dim, moma_prec=prec, queue_dtype=queue_dtype,
                             moma_fused=True, trace=[], overlap_teacher=overlap, graph_teacher=graph_teacher, graph_student=graph_student,
                             amp=amp)
    contrast = build_mem(opt)
    kd = CMO(opt)
    ms, mt, contrast, kd = ms.to(dev), mt.to(dev), contrast.to(dev), kd.to(dev)
    trainer = ContrastTrainer(opt)
    att_names = [n for n in ("atts", "atts_p", "atts_n", "atts_q", "atts_k", "atts_queue") if hasattr(kd, n)]
    trainable = nn.ModuleList([ms] + [getattr(kd, n) for n in att_names] + [kd.embed_s])
    if amp == "fp16":            # as train_student_moma.build_training / main_worker do for --amp fp16
        from moma_amd.train_student_moma import make_optimizer
        opt._grad_scaler = torch.amp.GradScaler("cuda", init_scale=scale0)
        opt.learning_rate, opt.momentum, opt.weight_decay = lr, 0.9, 1e-4
        optimizer = make_optimizer(trainable.parameters(), opt, dev)
        assert optimizer.defaults.get("fused") and getattr(optimizer, "_step_supports_amp_scaling", False)
    else:
        optimizer = torch.optim.SGD(trainable.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    student0 = torch.cat([p.detach().reshape(-1) for p in ms.parameters()]).clone()
    mods = nn.ModuleList([ms, kd.embed_s, kd.embed_t, mt])            # the CLI's module list with --head mlp
    crits = nn.ModuleList([nn.CrossEntropyLoss(), DistillKL(4.0), kd])
    gen = torch.Generator().manual_seed(3)
    data = [(torch.randn(B, 3, size, size, generator=gen), torch.randint(0, 10, (B,), generator=gen)) for _ in range(epochs * steps)]
    data.append((torch.randn(B - 3, 3, size, size, generator=gen), torch.randint(0, 10, (B - 3,), generator=gen)))   # ragged last batch
    if data_on_device:          # (as the CLI's synthetic loader hands them over: no host-blocking copy in the step, the host runs ahead)
        data = [(x.to(dev), y.to(dev)) for x, y in data]
    torch.manual_seed(99)                                              # the Shuffle-BN permutation stream (host generator)
    for ep in range(epochs):
        loader = data[ep * steps:(ep + 1) * steps] + ([data[-1]] if ep == epochs - 1 else [])
        if sync_tail:           # (diagnostics: a device synchronisation + a progress line in front of each of the epoch's last steps)
            loader = _SyncTail(loader, sync_tail, ep + 1)
        train_distill_moma(ep + 1, loader, mods, crits, trainer, contrast, optimizer, opt)
        if validate:            # as the CLI does between two epochs: every module in eval mode, the student forward without autocast
            from moma_amd.helper.loops_moma import validate_distill
            opt.n_cls = 10
            validate_distill(data[:2], mods, crits[0], opt, prefix="Val")
    torch.cuda.synchronize()
    sg = getattr(trainer, "_step_graphs", None)
    return dict(loss=torch.stack([t[0] for t in opt.trace]).cpu().numpy(), loss_kd=torch.stack([t[2] for t in opt.trace]).cpu().numpy(),
                index=[t[1] for t in opt.trace], memory=contrast.memory.float().cpu().numpy(), replays=0 if sg is None else sg.replays, ngraphs=0 if sg is None else len(sg.graphs),
                student={k: v.float().cpu().numpy() for k, v in ms.state_dict().items()},
                teacher={k: v.float().cpu().numpy() for k, v in mt.state_dict().items()},
                atts_q=getattr(kd, att_names[0]).proj.weight.detach().cpu().numpy(), next_perm=torch.randperm(16).tolist(),
                delta=(torch.cat([p.detach().reshape(-1) for p in ms.parameters()]) - student0).double().cpu().numpy(),
                scale=None if amp != "fp16" else float(opt._grad_scaler.get_scale()),
                atts_k_grad_none=(not hasattr(kd, "atts_queue")) or all(p.grad is None for p in kd.atts_k.parameters()),
                grads_attached=all(p.grad is not None for p in ms.parameters()))


@pytest.mark.parametrize("model,overlap,prec,queue_dtype,amp", [
    ("resnet8", True, "fp32", "fp32", None),            # exact-fp32 kernels (staged K1, one-pass fp32 K2), two streams
    ("resnet8", False, "bf16", "bf16", None),           # bf16 policy, everything on one stream
    ("resnet8", True, "bf16", "fp32", None),            # fp32 queue + bf16 mirror (K3 writes both)
    ("effiB0", True, "bf16", "bf16", "bf16"),           # the benchmark's configuration in small: BN / depthwise / SE helper kernels,
])                                                      # host-side batch counters, the bf16 weight cache
def test_step_graphs_equal_the_eager_loop(model, overlap, prec, queue_dtype, amp):
    """Two epochs + a ragged last batch with the step served from HIP graphs (from the 5th step on: the first step of an epoch
    -- teacher in eval mode -- and the ragged batch stay eager, so eager and replayed steps interleave) against the same run
    issued launch by launch: per-step loss and loss_kd, the queue pointer after every step (exact), the final queue, student,
    EMA teacher (incl. BatchNorm batch counters: exact) and attention weights, and the host generator's state afterwards
    (same number of permutations drawn)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    size, B, K, d, lr = (64, 16, 1024, 128, 2e-4) if model == "effiB0" else (32, 8, 256, 64, 0.02)
    a = _run(True, model, overlap, prec, queue_dtype, amp, B=B, K=K, d=d, size=size, lr=lr)
    b = _run(False, model, overlap, prec, queue_dtype, amp, B=B, K=K, d=d, size=size, lr=lr)
    assert a["replays"] == 3 + 6 and b["replays"] == 0          # epoch 1: steps 5-7 (capture at the 5th), epoch 2: steps 2-7
    assert a["index"] == b["index"] and a["index"][-1] == (14 * B + B - 3) % K
    assert a["next_perm"] == b["next_perm"]
    assert a["atts_k_grad_none"] and b["atts_k_grad_none"] and a["grads_attached"]
    _same(a, b)
    assert all(a["student"][name] == 15 for name in a["student"] if "num_batches_tracked" in name)


@pytest.mark.parametrize("prec,queue_dtype", [("bf16", "bf16"), ("fp32", "fp32")])
def test_the_eager_loop_repeats_bit_for_bit(prec, queue_dtype):
    """The premise of the exact comparisons in this file: under MIOpen's deterministic algorithms two runs of the loop in one process
    are bit-identical (no atomics anywhere in this library since round 6; fp32 at d = 64 takes the staged K2 path whose split-K
    gradient used them), with the batches resident on the device as the CLI's loader hands them over (the host runs ahead)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a = _run(False, "resnet8", True, prec, queue_dtype, None, data_on_device=True)
    b = _run(False, "resnet8", True, prec, queue_dtype, None, data_on_device=True)
    c = _run(True, "resnet8", True, prec, queue_dtype, None, data_on_device=True)
    _same(a, b)
    _same(c, b)
    assert c["replays"] == 9 and a["index"] == b["index"] == c["index"]


@pytest.mark.parametrize("model,amp,kw", [("resnet8", None, {}), ("effiB0", "bf16", dict(B=16, K=1024, d=128, size=64, lr=0.02))])
def test_execution_modes_change_no_bit(model, amp, kw):
    """Where the launches come from and which stream carries them must not change a single bit: the teacher side on its own stream
    or on the main one, the step graph-served or eager, the teacher's forward graph-served or not, the batches copied from the host
    or resident on the device (the host then runs ahead).  A difference would be a race or a dependence on issue order
    (scripts/diag_equivalences.py runs the longer form of this check; in round 6 it found the hang of
    test_an_eager_step_behind_replays_in_flight_finishes)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ref = _run(False, model, False, "bf16", "bf16", amp, **kw)                               # eager, one stream, host batches
    for graph, overlap, dev, gt in ((False, True, False, True), (True, False, False, True), (True, True, True, True), (True, True, True, False)):
        got = _run(graph, model, overlap, "bf16", "bf16", amp, data_on_device=dev, graph_teacher=gt, **kw)
        _same(got, ref)
        assert got["index"] == ref["index"] and got["next_perm"] == ref["next_perm"]


@pytest.mark.parametrize("attn", ["self_mix", "self_nomix"])
def test_step_graphs_serve_the_attention_in_shuffle_variants(attn):
    """--attn self_mix / self_nomix (learning/contrast_trainer.py:_shuffle_bn_attn, reference :135-187: the attention applied in
    front of the un-shuffle, over [q ; k] or per side) served from HIP graphs since round 6: the key encoding runs inside g_query
    behind the student's forward (it needs the student's query), the permutation comes through the same static feed, K2 packs q
    itself (no producer-side packed image).  Against the eager loop: per-step losses, pointer, queue, weights, generator state."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a = _run(True, "resnet8", True, "bf16", "bf16", None, attn=attn)
    b = _run(False, "resnet8", True, "bf16", "bf16", None, attn=attn)
    assert a["replays"] == 3 + 6 and b["replays"] == 0 and a["ngraphs"] == 1
    assert a["index"] == b["index"] and a["next_perm"] == b["next_perm"] and a["grads_attached"]
    _same(a, b)


@pytest.mark.parametrize("attn,prec", [("qk", "bf16"), ("self_qk", "fp32"), ("all", "bf16"), ("dual", "fp32"), ("self", "bf16")])
def test_step_graphs_serve_the_cross_attention_memory(attn, prec):
    """--mem MoCoAtt (reference MoMA/mem_moco.py:103-161; the loop path of SURVEY 8f n1) served from HIP graphs since round 6: the
    memory's cross-attention variant, the logits against a snapshot of the queue and their CrossEntropy are captured with the query
    side (MoCoAtt.forward_logits); only the enqueue -- its ring pointer is a host integer -- is issued between the graphs.  Against
    the eager loop (MoCoAtt.forward): per-step losses, pointer after every step, final queue, weights, generator state."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a = _run(True, "resnet8", True, prec, "fp32", None, attn=attn, mem="MoCoAtt")
    b = _run(False, "resnet8", True, prec, "fp32", None, attn=attn, mem="MoCoAtt")
    assert a["replays"] == 3 + 6 and b["replays"] == 0 and a["ngraphs"] == 1
    assert a["index"] == b["index"] and a["index"][-1] == (14 * 8 + 8 - 3) % 256
    assert a["next_perm"] == b["next_perm"] and a["grads_attached"]
    _same(a, b)


@pytest.mark.parametrize("scale0", [2.0 ** 10, 2.0 ** 22])
def test_step_graphs_with_fp16_and_a_grad_scaler(scale0):
    """--amp fp16 (BASELINE configs[4]): fp16 autocast + GradScaler with the step served from HIP graphs -- the captured backward
    starts from loss * scale (the scaler's DEVICE tensor, read at replay time), `scaler.step()` hands scale and found-inf to the
    fused SGD as device tensors and `scaler.update()` adjusts the scale on the device: no host read-back anywhere, replays and
    eager steps interleave.  Against the same run issued launch by launch: per-step losses, pointer, final scale (exact: the
    same sequence of finite / overflowed steps).  scale0 = 2^22 overflows fp16 gradients for the first steps: those updates are
    skipped INSIDE the optimizer kernel and the scale backs off -- on both paths alike, also across the capture."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a = _run(True, "resnet8", True, "bf16", "bf16", "fp16", scale0=scale0, lr=2e-4)
    b = _run(False, "resnet8", True, "bf16", "bf16", "fp16", scale0=scale0, lr=2e-4)
    assert a["replays"] == 3 + 6 and b["replays"] == 0 and a["ngraphs"] == 1
    assert a["index"] == b["index"] and a["next_perm"] == b["next_perm"]
    assert a["scale"] == b["scale"] and (a["scale"] < scale0 if scale0 > 2.0 ** 20 else a["scale"] == scale0), (a["scale"], b["scale"])
    assert np.isfinite(a["loss"]).all() and np.isfinite(b["loss"]).all()
    _same(a, b)
    assert all(a["student"][name] == 15 for name in a["student"] if "num_batches_tracked" in name)


def test_an_eager_step_behind_replays_in_flight_finishes():
    """Round 6: with the teacher side on its own stream and the batches already on the device (nothing in a step blocks the host, so
    it runs up to three replayed steps ahead), the EAGER step of a ragged last batch issued behind replays still in flight never
    finished -- the epoch's closing read-back waited for good (second run of a process, EfficientNet pair with bf16 autocast:
    reproducible; not with one stream, not with a synchronisation in front of the switch).  StepGraphs now drains the device where
    the loop switches between replayed and eager steps.  Runs in a child process under a watchdog (a regression would hang, not
    fail): three consecutive graph-served runs must finish."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HANG="1", VAL="0", DEVDATA="1", ONLY="effiB0", MODE_LIMIT_S="90")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "diag_equivalences.py")], capture_output=True, text=True, timeout=600,
                       cwd=root, env=env)
    assert r.returncode == 0 and r.stdout.count("finished, last loss") == 3, r.stdout[-1500:] + r.stderr[-3000:]


def test_step_graphs_follow_a_replaced_queue_and_the_optimizer():
    """What is NOT frozen in the graphs: the optimizer (eager: a changed learning rate takes effect at once -- lr = 0 leaves
    the weights where they are while the replayed backward still fills the gradients) and the queue storage (K2 / K3 run
    between the graphs on the live tensors; a replaced `memory` gets a capture of its own, never a stale address)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.backbones import model_dict
    from moma_amd.MoMA.mem_moco import build_mem
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.distiller_zoo import DistillKL
    torch.backends.cudnn.benchmark = False
    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    B, K, d = 8, 128, 64
    opt = argparse.Namespace(distill="moma", head="mlp", feat_dim=d, attn="self", mem="MoCo", nce_k=K, nce_t=0.15, alpha=0.99,
                             cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=0, multiprocessing_distributed=False, print_freq=1000,
                             batch_size=B, rank=0, world_size=1, s_dim=64, t_dim=64, moma_prec="bf16", queue_dtype="bf16",
                             moma_fused=True, trace=[], overlap_teacher=True)
    ms, mt = model_dict["resnet8"](num_classes=10).to(dev), model_dict["resnet8"](num_classes=10).to(dev)
    contrast, kd = build_mem(opt).to(dev), CMO(opt).to(dev)
    trainer = ContrastTrainer(opt)
    trainable = nn.ModuleList([ms, kd.atts_q, kd.atts_k, kd.atts_queue, kd.embed_s])
    optimizer = torch.optim.SGD(trainable.parameters(), lr=0.02, momentum=0.0, weight_decay=0.0)
    mods, crits = nn.ModuleList([ms, mt]), nn.ModuleList([nn.CrossEntropyLoss(), DistillKL(4.0), kd])
    gen = torch.Generator().manual_seed(3)
    batch = lambda: (torch.randn(B, 3, 32, 32, generator=gen), torch.randint(0, 10, (B,), generator=gen))
    train_distill_moma(1, [batch() for _ in range(7)], mods, crits, trainer, contrast, optimizer, opt)
    sg = trainer._step_graphs
    assert sg.replays == 3 and len(sg.graphs) == 1
    # lr = 0: replayed steps, weights frozen, gradients still produced
    w0 = [p.detach().clone() for p in ms.parameters()]
    for g in optimizer.param_groups:
        g["lr"] = 0.0
    train_distill_moma(2, [batch() for _ in range(4)], mods, crits, trainer, contrast, optimizer, opt)
    assert sg.replays == 3 + 3 and all(torch.equal(a, b) for a, b in zip(w0, ms.parameters()))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0 for p in ms.parameters())
    # a replaced queue: same shapes, another tensor -> the old capture is not used for it; after the warm-up a new one serves
    old_ptr = contrast.memory.data_ptr()
    retired = [contrast.memory]             # (kept alive: a freed queue's block may be handed out again for a later one, and a queue
    #                                          at a captured address IS served by that capture -- rightly, but not what is tested here)
    contrast.memory = torch.nn.functional.normalize(torch.randn(K, d, device=dev)).to(torch.bfloat16)
    assert contrast.memory.data_ptr() != old_ptr
    before = contrast.memory.clone()
    idx0 = contrast.index
    opt.trace.clear()
    train_distill_moma(3, [batch() for _ in range(7)], mods, crits, trainer, contrast, optimizer, opt)
    assert len(sg.graphs) == 2 and sg.replays == 6 + 3
    assert [t[1] for t in opt.trace] == [(idx0 + (i + 1) * B) % K for i in range(7)]
    rows = [(idx0 + i) % K for i in range(7 * B)]
    untouched = [r for r in range(K) if r not in set(rows)]
    assert torch.equal(contrast.memory[untouched], before[untouched]) and not torch.equal(contrast.memory[rows], before[rows])
    # a THIRD storage: the runner is at capacity (max_graphs = 2) -- the variant unused longest (the first queue's, which can never
    # match again) makes room instead of leaving every later variant eager for good
    retired.append(contrast.memory)
    contrast.memory = torch.nn.functional.normalize(torch.randn(K, d, device=dev)).to(torch.bfloat16)
    assert contrast.memory.data_ptr() not in [t.data_ptr() for t in retired]
    idx0 = contrast.index
    opt.trace.clear()
    train_distill_moma(4, [batch() for _ in range(7)], mods, crits, trainer, contrast, optimizer, opt)
    assert len(sg.graphs) == 2 and sg.replays == 9 + 3
    assert [t[1] for t in opt.trace] == [(idx0 + (i + 1) * B) % K for i in range(7)]
    assert all(torch.isfinite(t[0]) for t in opt.trace)


# ---- helper/graphs.py: the capture hazards of round 3 ------------------------------------------------------
def _small_net(dev):
    torch.manual_seed(5)
    from moma_amd.backbones.resnet_cifar import resnet8
    return resnet8(num_classes=10).to(dev).eval()


def test_graphed_inference_is_keyed_by_stream():
    """A graph captured for one stream is never replayed on another (library workspaces belong to the capturing stream: round 3
    saw a variant primed on the main stream and replayed on the side stream next to the student forward return garbage).  Prime on
    stream A, call on stream B: B must serve its own eager calls and capture its OWN graph; results equal the plain module."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.helper.graphs import GraphedInference
    dev = torch.device("cuda", 0)
    net = _small_net(dev)
    x = torch.randn(8, 3, 32, 32, device=dev)
    with torch.no_grad():
        ref_feats, ref_logits = net(x, is_feat=True)
    g = GraphedInference(net, warmup=2)
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(a):
        assert g.prime(x, is_feat=True)
    torch.cuda.synchronize()
    assert len(g._graphs) == 1
    with torch.cuda.stream(b), torch.no_grad():
        outs = [g(x, is_feat=True) for _ in range(4)]          # 2 eager + capture + replay, all on b
    torch.cuda.synchronize()
    assert len(g._graphs) == 2                                 # b got its own graph
    keys = list(g._graphs)
    assert keys[0][-1] != keys[1][-1] and {keys[0][-1], keys[1][-1]} == {a.cuda_stream, b.cuda_stream}
    for feats, logits in outs:
        assert torch.allclose(logits, ref_logits, rtol=1e-4, atol=1e-4) and torch.allclose(feats[-1], ref_feats[-1], rtol=1e-4, atol=1e-4)


def test_graphed_inference_capture_inside_a_live_autocast_context():
    """Several calls inside ONE autocast context share autocast's weight-cast cache; a capture that reused a cast cached by an
    earlier eager call would hold no cast kernel and read a low-precision weight that is freed when the context exits (round 3:
    NaN, then a memory fault, with the ViT-B teacher under fp16).  Eager calls, the capture and replays all inside one
    context, then replays after the context has exited and after the weights CHANGED: the replay must follow the weights."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.helper.graphs import GraphedInference
    from moma_amd.backbones import model_dict
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    net = model_dict["vit_tiny_patch16_224"](num_classes=5).to(dev).eval()       # Linear layers: autocast casts their weights
    x = torch.randn(4, 3, 64, 64, device=dev)
    g = GraphedInference(net, warmup=2)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        ref = net(x, is_feat=True)[1].float()
        outs = [g(x, is_feat=True)[1].float() for _ in range(4)]            # eager, eager, capture + replay, replay
    torch.cuda.synchronize()
    assert len(g._graphs) == 1
    for o in outs:
        assert torch.isfinite(o).all() and torch.allclose(o, ref, rtol=2e-2, atol=2e-2)
    # churn the allocator (a freed cached cast would be overwritten by now), change the weights, replay in a NEW context
    junk = [torch.randn(1 << 20, device=dev) for _ in range(8)]
    del junk
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        ref2 = net(x, is_feat=True)[1].float()
        out2 = g(x, is_feat=True)[1].float()
    torch.cuda.synchronize()
    assert len(g._graphs) == 1
    assert torch.isfinite(out2).all() and torch.allclose(out2, ref2, rtol=2e-2, atol=2e-2)
    assert not torch.allclose(out2, ref, rtol=1e-3, atol=1e-3)             # (it did follow the new weights)


def test_runtime_replays_captured_memsets_after_eager_work():
    """The hazard behind moma_amd/hip_env.py (ROCm 7.2 graph packet capture: a captured hipMemsetAsync node stops working once
    eager kernels ran between replays -- garbage from every ATen column sum behind it).  The package's import must have switched the
    packet capture off before the first HIP call, and the self-test every capture of this package consults must pass on this box."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import os
    import moma_amd  # noqa: F401
    from moma_amd.helper import graphs
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") == "0"
    graphs._REPLAY_SAFE.clear()
    assert graphs.replay_is_safe(torch.device("cuda", 0))
    assert graphs._REPLAY_SAFE == {0: True}


def test_a_runtime_that_fails_the_replay_self_test_keeps_the_eager_loop():
    """helper/graphs.py:replay_is_safe() False (a process whose HIP runtime was initialised with graph packet capture on) -> no
    capture anywhere: the step graphs and the teacher's GraphedInference both stay eager, the loop runs and equals the eager run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.helper import graphs
    saved = dict(graphs._REPLAY_SAFE)
    try:
        graphs._REPLAY_SAFE.clear()
        graphs._REPLAY_SAFE[0] = False
        a = _run(True, "resnet8", True, "bf16", "bf16", None)
    finally:
        graphs._REPLAY_SAFE.clear()
        graphs._REPLAY_SAFE.update(saved)
    b = _run(False, "resnet8", True, "bf16", "bf16", None)
    assert a["replays"] == 0 and b["replays"] == 0
    assert a["index"] == b["index"] and a["next_perm"] == b["next_perm"]
    _same(a, b)


def test_the_once_per_epoch_variant_never_earns_a_capture():
    """The step that opens every epoch (teacher in eval mode, reference helper/loops_moma.py:227) is a variant of its own: seen
    once per epoch, it must stay eager however many epochs pass (a capture costs a second set of graph pools -- 13 GB reserved at
    the bench shape).  Six epochs: one variant captured, five steps replayed in each epoch after the first."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a = _run(True, "resnet8", True, "bf16", "bf16", None, epochs=6, steps=6)
    from moma_amd.helper import step_graph  # noqa: F401
    assert a["ngraphs"] == 1
    assert a["replays"] == 2 + 5 * 5           # epoch 1: steps 5-6; epochs 2-6: steps 2-6 (the ragged batch and each first step eager)
    assert np.isfinite(a["loss"]).all()


def test_a_moved_parameter_is_never_replayed_through_its_old_address():
    """The captured graphs hold the addresses of parameters and buffers.  A tensor that moves between two steps (here: one student
    weight re-created with the same values) must make the next steps another variant -- eager, then captured anew -- never a replay
    through the stale pointer: the run continues exactly as the eager loop does."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.backbones import model_dict
    from moma_amd.MoMA.mem_moco import build_mem
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.distiller_zoo import DistillKL

    def run(graph_student):
        torch.backends.cudnn.benchmark = False
        dev = torch.device("cuda", 0)
        torch.manual_seed(1)
        B, K, d = 8, 128, 64
        opt = argparse.Namespace(distill="moma", head="mlp", feat_dim=d, attn="self", mem="MoCo", nce_k=K, nce_t=0.15, alpha=0.99,
                                 cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=0, multiprocessing_distributed=False, print_freq=1000,
                                 batch_size=B, rank=0, world_size=1, s_dim=64, t_dim=64, moma_prec="bf16", queue_dtype="bf16",
                                 moma_fused=True, trace=[], overlap_teacher=True, graph_student=graph_student)
        ms, mt = model_dict["resnet8"](num_classes=10).to(dev), model_dict["resnet8"](num_classes=10).to(dev)
        contrast, kd = build_mem(opt).to(dev), CMO(opt).to(dev)
        trainer = ContrastTrainer(opt)
        trainable = nn.ModuleList([ms, kd.atts_q, kd.atts_k, kd.atts_queue, kd.embed_s])
        optimizer = torch.optim.SGD(trainable.parameters(), lr=0.02, momentum=0.0, weight_decay=0.0)
        mods, crits = nn.ModuleList([ms, kd.embed_s, kd.embed_t, mt]), nn.ModuleList([nn.CrossEntropyLoss(), DistillKL(4.0), kd])
        gen = torch.Generator().manual_seed(3)
        batch = lambda: (torch.randn(B, 3, 32, 32, generator=gen), torch.randint(0, 10, (B,), generator=gen))
        torch.manual_seed(5)
        train_distill_moma(1, [batch() for _ in range(7)], mods, crits, trainer, contrast, optimizer, opt)
        # move one weight: a new tensor with the same values, registered under the same name (and known to the optimizer)
        name, old = next((n, p) for n, p in ms.named_parameters() if p.dim() == 4)
        new = nn.Parameter(old.detach().clone())
        holder = ms
        for part in name.split(".")[:-1]:
            holder = getattr(holder, part)
        setattr(holder, name.split(".")[-1], new)
        for g in optimizer.param_groups:
            g["params"] = [new if p is old else p for p in g["params"]]
        del old
        train_distill_moma(2, [batch() for _ in range(7)], mods, crits, trainer, contrast, optimizer, opt)
        sg = getattr(trainer, "_step_graphs", None)
        return (torch.stack([t[0] for t in opt.trace]).cpu().numpy(), 0 if sg is None else sg.replays,
                0 if sg is None else len(sg.graphs), [t[1] for t in opt.trace])
    lg, replays, ngraphs, idx_g = run(True)
    le, none, _, idx_e = run(False)
    assert replays == 3 + 3 and ngraphs == 2 and none == 0       # epoch 1: steps 5-7; epoch 2: warm-up of the new variant, then steps 5-7
    assert idx_g == idx_e
    np.testing.assert_allclose(lg, le, rtol=2e-4, atol=2e-4)


def test_graphed_inference_makes_room_at_capacity():
    """At `max_graphs` variants the one unused longest is dropped for a new one (a variant keyed on weights that moved can never
    match again): five input shapes through a wrapper that holds two graphs -- every shape ends up served from a graph, results
    equal the plain module."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.helper.graphs import GraphedInference
    dev = torch.device("cuda", 0)
    net = _small_net(dev)
    g = GraphedInference(net, warmup=1, max_graphs=2)
    with torch.no_grad():
        for bs in (2, 3, 4, 5, 6):
            x = torch.randn(bs, 3, 32, 32, device=dev)
            ref = net(x, is_feat=True)[1]
            outs = [g(x, is_feat=True)[1] for _ in range(3)]        # eager, capture + replay, replay
            assert len(g._graphs) <= 2 and g._key(x, True) in g._graphs
            for o in outs:
                assert torch.allclose(o, ref, rtol=1e-4, atol=1e-4)
    torch.cuda.synchronize()
