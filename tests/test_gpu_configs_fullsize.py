"""BASELINE configs[2] and configs[4] (single-GPU form) at FULL size, three steps of the real loop each, with the KD term of the
second step checked against the numpy oracle at the real shapes (these configurations existed only as
builder-run bench lines; their -m gpu tests were 64-pixel smokes without a numeric check).

    configs[2]  ViT-S student + teacher, 224 x 224, B = 256, --head None -> d = 384, 8 attention heads (head dim 48), bf16
    configs[4]  ViT-B teacher -> ResNet-50 student, 512 x 512, B = 64, --head mlp (2048 / 768 -> 512), fp16 autocast + GradScaler,
                teacher frozen (the architectures differ: SURVEY Q4)

The loop is bench.py's (same option namespace, build_training, ContrastTrainer, train_distill_moma).  On the second step the
tensors that enter K1 (the query side: f_s, the attention weights) and K2 (q, k, the pre-enqueue queue) are snapshotted and
    K1   atts_q(f_s)                 vs  oracle.attention_fwd            (MoMA/criterion_moco_att.py:153-167)     2e-2 of max
    K2   loss_kd                     vs  oracle.compute_logit -> infonce_loss  (MoMA/mem_moco.py:29-49 + CrossEntropy)  1e-3
         d loss_kd / d q (autograd)  vs  oracle.infonce_grad                                                     2e-2 of max
    K3   the rows the step enqueued sit at [index, index + B) of the queue, rounded to its bf16 storage; pointer exact
are compared -- the backbones (ViT / ResNet on hipBLASLt / MIOpen) only have to produce finite features."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import moma_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _three_steps(argv):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from moma_amd.train_student_moma import build_training
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.dataset.synthetic import SyntheticLoader

    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = False
    opt = bench.make_opt(a, 0, 1)
    opt.trace = []
    torch.manual_seed(12345)
    model_s, model_t, module_list, criterion_list, _tr, contrast, optimizer = build_training(opt, dev)
    trainer = ContrastTrainer(opt)
    scaler = None
    if opt.amp == "fp16":
        scaler = opt._grad_scaler = torch.amp.GradScaler("cuda")
    kd = criterion_list[2]
    rec, step = {}, [0]
    att_forward, fused = kd.atts_q.forward, contrast.forward_fused

    def att(x, qpack=None):
        y = att_forward(x, qpack=qpack)
        if step[0] == 1:
            rec.update(att_x=x.detach().float().clone(), att_y=y.detach().clone(),
                       att_w={k: v.detach().clone() for k, v in kd.atts_q.state_dict().items()})
        return y

    def ff(q, k, all_k=None, qpack=None):
        take = step[0] == 1
        if take:
            rec.update(q=q.detach().clone(), k=k.detach().clone(), all_k=(k if all_k is None else all_k).detach().clone(),
                       queue=contrast.memory.detach().clone(), index=contrast.index,
                       scale=1.0 if scaler is None else float(scaler.get_scale()))
            q.register_hook(lambda g: rec.__setitem__("dq", g.detach().clone()))
        out = fused(q, k, all_k, **({"qpack": qpack} if qpack is not None else {}))
        if take:
            rec["loss_kd"] = out[0].detach().clone()
        step[0] += 1
        return out
    kd.atts_q.forward, contrast.forward_fused = att, ff
    loader = SyntheticLoader(3, a.batch_size, a.image_size, a.n_cls, 12345, dev)
    train_distill_moma(1, loader, module_list, criterion_list, trainer, contrast, optimizer, opt)
    torch.cuda.synchronize()
    return a, opt, contrast, kd, rec


def _check_kd_term(a, opt, contrast, kd, rec, heads):
    B, d, K, T = a.batch_size, contrast.memory.shape[1], a.nce_k, 0.15
    N = lambda t: t.detach().float().cpu().numpy()
    assert rec["q"].shape == (B, d) and rec["queue"].shape == (K, d) and rec["index"] == B % K
    losses = [float(t[0]) for t in opt.trace]
    assert len(losses) == 3 and all(np.isfinite(losses)) and [t[1] for t in opt.trace] == [(i + 1) * B % K for i in range(3)]
    assert contrast.index == 3 * B % K                                             # pointer: exact, host integer
    # K1, query side (bf16 policy: 2e-2 of the output's range)
    w = rec["att_w"]
    y_ref = O.attention_fwd(N(rec["att_x"]), N(w["qkv.weight"]), N(w["qkv.bias"]), N(w["proj.weight"]), N(w["proj.bias"]), heads)
    assert np.abs(N(rec["att_y"]) - y_ref).max() < 2e-2 * np.abs(y_ref).max()
    assert torch.equal(rec["att_y"], rec["q"])                                     # K2 took exactly what K1 produced
    # K2: loss within the north star's 1e-3, gradient w.r.t. q within 2e-2 of its range; the oracle reads the queue's stored values
    queue = N(rec["queue"])
    ref = O.infonce_loss(O.compute_logit(N(rec["q"]), N(rec["k"]), queue, T, dtype=np.float64))
    assert abs(float(rec["loss_kd"]) - ref["loss"]) < 1e-3 * max(1.0, abs(ref["loss"])), (float(rec["loss_kd"]), ref["loss"])
    dq_ref = O.infonce_grad(N(rec["q"]), N(rec["k"]), queue, T)                    # d mean(loss_rows) / dq
    dq = N(rec["dq"]).astype(np.float64) / (rec["scale"] * opt.beta)               # autograd: beta * GradScaler's factor on top
    assert np.abs(dq - dq_ref).max() < 2e-2 * np.abs(dq_ref).max()
    # K3: step 2's rows in the slots [index, index + B) (not yet overwritten: 3 B < K), rounded to the queue's storage
    rows = torch.arange(rec["index"], rec["index"] + B, device=contrast.memory.device) % K
    assert torch.equal(contrast.memory[rows], rec["all_k"].to(contrast.memory.dtype))
    untouched = torch.arange(3 * B, K, device=contrast.memory.device)
    assert torch.equal(contrast.memory[untouched], rec["queue"][untouched])


def test_config1_effnet_b0_pair_full_size():
    """BASELINE configs[1] ITSELF, the configuration the headline metric is quoted on and bench.py times: EfficientNet-B0 student +
    teacher, 224 x 224, per-GPU batch 256, queue K = 65536 x d = 512 (--head mlp), 4-head attention-KD path, bf16 -- three steps of
    the loop with the second step's K1 -> K2 -> K3 checked against the numpy oracle (round 4 carried this config by bench.py and by
    G5c, the same KD shapes on a resnet8 backbone, only)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a, opt, contrast, kd, rec = _three_steps(["--no_cpu_baseline"])
    assert (a.model, a.batch_size, a.image_size, a.nce_k, a.head, a.amp) == ("effiB0", 256, 224, 65536, "mlp", "bf16")
    assert contrast.memory.shape == (65536, 512) and contrast.memory.dtype == torch.bfloat16 and kd.atts_q.num_heads == 4
    assert opt.s_dim == 1280 and opt.t_dim == 1280
    _check_kd_term(a, opt, contrast, kd, rec, heads=4)


def test_config2_vit_small_pair_full_size():
    """BASELINE configs[2] as named: ViT-Small student + teacher, 224 x 224, per-GPU batch 256, 8-head attention-KD path
    (d = 384 -> head dim 48 on the K1 fast path, one-pass K2 at d = 384), bf16."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a, opt, contrast, kd, rec = _three_steps(["--model", "vit_small_patch16_224", "--head", "None", "--num_heads", "8",
                                              "--batch_size", "256", "--image_size", "224", "--no_cpu_baseline"])
    assert contrast.memory.shape == (65536, 384) and kd.atts_q.num_heads == 8
    _check_kd_term(a, opt, contrast, kd, rec, heads=8)


def test_config4_vit_base_to_resnet50_full_size_single_gpu_form():
    """BASELINE configs[4], one rank of it: ViT-Base teacher -> ResNet-50 student, 512 x 512, per-GPU batch 64, fp16 autocast +
    GradScaler, --head mlp (s_dim 2048, t_dim 768 -> d = 512), teacher frozen (cross-architecture)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a, opt, contrast, kd, rec = _three_steps(["--model", "ResNet50", "--model_t", "vit_base_patch16_224", "--image_size", "512",
                                              "--batch_size", "64", "--amp", "fp16", "--no_cpu_baseline"])
    assert contrast.memory.shape == (65536, 512) and opt.s_dim == 2048 and opt.t_dim == 768 and rec["scale"] > 1.0
    _check_kd_term(a, opt, contrast, kd, rec, heads=4)


def _losses(argv, steps_per_epoch, scaler_init=None):
    """per-step (loss, loss_kd) of bench.py's loop over several epochs, printing EVERY step (eager kernels + a read-back between
    the replays of the step graphs)"""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from moma_amd.train_student_moma import build_training
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.dataset.synthetic import SyntheticLoader

    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = False
    opt = bench.make_opt(a, 0, 1)
    opt.trace, opt.print_freq = [], 1
    torch.manual_seed(12345)
    model_s, model_t, module_list, criterion_list, _tr, contrast, optimizer = build_training(opt, dev)
    trainer = ContrastTrainer(opt)
    if opt.amp == "fp16":
        opt._grad_scaler = torch.amp.GradScaler("cuda", **({} if scaler_init is None else {"init_scale": scaler_init}))
    for ep, n in enumerate(steps_per_epoch):
        loader = SyntheticLoader(n, a.batch_size, a.image_size, a.n_cls, 12345 + ep, dev)
        train_distill_moma(ep, loader, module_list, criterion_list, trainer, contrast, optimizer, opt)
    torch.cuda.synchronize()
    sg = getattr(trainer, "_step_graphs", None)
    finite = all(bool(torch.isfinite(p).all()) for p in model_s.parameters())
    return (np.array([float(t[0]) for t in opt.trace]), np.array([float(t[2]) for t in opt.trace]),
            0 if sg is None else sg.replays, finite)


def test_config2_step_graphs_survive_eager_work_between_replays():
    """Round 4: with the ROCm runtime's graph packet capture on, a captured memset node (ATen zeroes the semaphores of its column
    reductions that way: the bias gradients of the ViT's wide Linear layers) stops working once eager kernels run between two
    replays -- the ViT-S pair of configs[2] went to NaN two steps after the loop's first print.  moma_amd/hip_env.py switches
    the packet capture off, helper/graphs.py:replay_is_safe() checks the live runtime before any capture.  Here: configs[2] at
    full size, two epochs, a print (device arithmetic + read-back) after EVERY step; graphs against the eager loop."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.helper.graphs import replay_is_safe
    assert replay_is_safe(torch.device("cuda", 0))            # (this box's runtime with the package's environment)
    argv = ["--model", "vit_small_patch16_224", "--head", "None", "--num_heads", "8", "--batch_size", "256", "--image_size", "224",
            "--learning_rate", "0.005", "--no_cpu_baseline"]
    lg, kg, replays, finite_g = _losses(argv, [5, 6])
    le, ke, none, finite_e = _losses(argv + ["--no_graph_student"], [5, 6])
    assert replays == 2 + 6 and none == 0                     # (the ViT has no BatchNorm: one variant, captured at the 4th step)
    assert finite_g and finite_e and np.isfinite(lg).all() and np.isfinite(kg).all()
    np.testing.assert_allclose(lg, le, rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(kg, ke, rtol=5e-3, atol=5e-3)


def test_config4_fp16_grad_scaler_step_graphs_against_the_eager_loop():
    """BASELINE configs[4] (single-GPU form) at full size with the step SERVED FROM HIP GRAPHS (round 5: fp16 + GradScaler joined
    the graphable configurations -- fused SGD, scale and found-inf flag stay on the device): two epochs, a print after every step,
    against the same run issued launch by launch.  Small learning rate so that the two trajectories stay comparable (fp16 autocast
    is not reproducible run to run), and the scaler starts at 2^12: at the default 2^16 the first step overflows, the scale settles
    at 2^15 -- and there the largest element of conv1's fp16 weight gradient sits at 0.92 - 0.99 of the fp16 maximum, so every few
    steps ONE element overflows and that step is skipped, on either path, in a different step from run to run
    (scripts/diag_cfg4_graph.py: graph-served and eager runs alike).  That is the GradScaler doing its job, and tests/
    test_gpu_step_graph.py covers skipped steps under graphs at a small size; here it would only make the comparison a lottery."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    argv = ["--model", "ResNet50", "--model_t", "vit_base_patch16_224", "--image_size", "512", "--batch_size", "64", "--amp", "fp16",
            "--learning_rate", "2e-4", "--no_cpu_baseline"]
    lg, kg, replays, finite_g = _losses(argv, [5, 6], scaler_init=4096.0)
    le, ke, none, finite_e = _losses(argv + ["--no_graph_student"], [5, 6], scaler_init=4096.0)
    assert replays >= 6 and none == 0, (replays, none)
    assert finite_g and finite_e and np.isfinite(lg).all() and np.isfinite(le).all(), (lg, le)
    np.testing.assert_allclose(lg, le, rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(kg, ke, rtol=5e-3, atol=5e-3)
