"""GPU parity of the whole MoMA step: the HIP-backed loop (moma_amd.helper.loops_moma.train_distill_moma with
MoCo / CMO / ContrastTrainer from moma_amd) against the trace captured from the reference loop itself
(tests/golden/g5_step_trace.npz).  fp32 policy; the backbone convolutions run on MIOpen instead of the CPU,
so per-step losses are compared at 2e-3 (north star: loss within 1e-3 of the reference holds on the KD term,
checked separately), the queue pointer bit-exactly."""
import argparse
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


@pytest.mark.parametrize("fixture,ci", [("g5_step_trace.npz", 0), ("g5_step_trace.npz", 1),
                                        ("g10_step_trace_heads.npz", 0), ("g10_step_trace_heads.npz", 1)])
@pytest.mark.parametrize("fused,overlap", [(True, False), (False, False), (True, True)])
def test_loop_matches_reference_trace(golden_dir, fixture, ci, fused, overlap):
    """10 steps of train_distill_moma against the trace captured from the reference (G5: heads None / mlp; G10: heads linear /
    mlp_byol, which the reference leaves untrained and -- mlp_byol -- with both BatchNorm1d layers in training mode): losses,
    queue pointer, final queue / weights.  overlap = the teacher / key side of every step on a second HIP stream (and, from the
    4th call on, the teacher forwards replayed from HIP graphs): scheduling only, the numbers must not move."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.backbones.resnet_cifar import resnet8
    from moma_amd.MoMA.mem_moco import build_mem
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.distiller_zoo import DistillKL

    torch.backends.cudnn.benchmark = False
    g = np.load(os.path.join(golden_dir, fixture))
    p = f"c{ci}_"
    head = str(g[p + "head"])
    feat_dim = 64 if head == "None" else 32
    opt = argparse.Namespace(distill="moma", head=head, feat_dim=feat_dim, attn="self", mem="MoCo", nce_k=64,
                             nce_t=0.15, alpha=0.999, cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=0,
                             multiprocessing_distributed=False, print_freq=1000, batch_size=8, rank=0,
                             world_size=1, s_dim=64, t_dim=64, moma_prec="fp32", moma_fused=fused, trace=[],
                             overlap_teacher=overlap)
    dev = torch.device("cuda", 0)
    ms, mt = resnet8(num_classes=100), resnet8(num_classes=100)
    ms.load_state_dict(_sd(g, p + "s.")); mt.load_state_dict(_sd(g, p + "t."))
    contrast = build_mem(opt)
    contrast.memory.copy_(torch.from_numpy(g[p + "memory0"]))
    kd = CMO(opt)
    kd.load_state_dict(_sd(g, p + "kd."))
    ms, mt, contrast, kd = ms.to(dev), mt.to(dev), contrast.to(dev), kd.to(dev)
    trainer = ContrastTrainer(opt)
    trainable = nn.ModuleList([ms, kd.atts_q, kd.atts_k, kd.atts_queue])
    if head == "mlp":
        trainable.append(kd.embed_s)
    optimizer = torch.optim.SGD(trainable.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    mods = nn.ModuleList([ms, mt])
    crits = nn.ModuleList([nn.CrossEntropyLoss(), DistillKL(opt.kd_T), kd])

    gen = torch.Generator().manual_seed(int(g[p + "data_seed"]))
    images = torch.randn(10, 8, 3, 32, 32, generator=gen)
    labels = torch.randint(0, 100, (10, 8), generator=gen)
    torch.manual_seed(int(g[p + "loop_seed"]))      # same randperm stream as the reference run
    for ep in range(2):
        loader = [(images[ep * 5 + i], labels[ep * 5 + i]) for i in range(5)]
        train_distill_moma(ep + 1, loader, mods, crits, trainer, contrast, optimizer, opt)
    losses = [float(t[0]) for t in opt.trace]
    idxs = [t[1] for t in opt.trace]
    # the trace records the pointer BEFORE the step's enqueue is visible? no: after forward_fused -> after enqueue
    assert idxs == [int(v) for v in g[p + "index"]]
    np.testing.assert_allclose(losses, g[p + "loss"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(contrast.memory.float().cpu().numpy(), g[p + "memory_final"], rtol=0, atol=2e-3)
    assert all(q.grad is None for q in kd.atts_k.parameters())
    assert all(q.grad is None for q in kd.atts_queue.parameters())
    assert np.array_equal(kd.atts_k.proj.weight.detach().cpu().numpy(), g[p + "kd_final.atts_k.proj.weight"])
    np.testing.assert_allclose(mt.fc.weight.detach().cpu().numpy(), g[p + "t_final.fc.weight"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(kd.atts_q.proj.weight.detach().cpu().numpy(), g[p + "kd_final.atts_q.proj.weight"],
                               rtol=0, atol=2e-3)
    if fixture.startswith("g10"):
        final = _sd(g, p + "kd_final.")
        for name, t in kd.state_dict().items():
            if not name.startswith("embed_"):
                continue
            if "running" in name or "num_batches" in name:                   # BatchNorm1d statistics moved like the reference's
                np.testing.assert_allclose(t.float().cpu().numpy(), final[name].float().numpy(), rtol=0, atol=2e-3)
            else:                                                            # weights: never registered with the optimizer
                assert np.array_equal(t.cpu().numpy(), final[name].numpy()), name


@pytest.mark.parametrize("ci", [0, 1, 2])
@pytest.mark.parametrize("queue_dtype,overlap", [("fp32", True), ("bf16", True), ("bf16", False)])
def test_loop_bf16_policy_matches_reference_trace_big_queue(golden_dir, ci, queue_dtype, overlap):
    """The configuration the benchmark times, tied to the reference at LOOP level (G5b: K = 65536, --head mlp, d = 128 /
    512): moma_prec = bf16 -> the one-pass K2 (infonce_flash_kernel), the fused K1 core forward + backward, K3 on the queue
    (fp32 storage + bf16 mirror, or bf16 storage), K4; teacher side on the second stream and replayed from HIP graphs when
    overlap is on.  Backbone (resnet8) stays fp32 so the comparison isolates the KD kernels.
    Tolerances: per-step loss_kd within 1e-3 relative of the reference (north star; the lr = 0.05 cases collapse to
    loss_kd = 0 by step 4, there 1e-3 absolute), total loss 5e-3, pointer exact, final queue rows equal after bf16 rounding."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.g5b_util import K_BIG, batches, big_queue, fill_attention_, sd
    from moma_amd.backbones.resnet_cifar import resnet8
    from moma_amd.MoMA.mem_moco import build_mem
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.distiller_zoo import DistillKL
    from moma_amd import _lib

    torch.backends.cudnn.benchmark = False
    g = np.load(os.path.join(golden_dir, "g5b_step_trace_big.npz"))
    p = f"c{ci}_"
    d = int(g[p + "feat_dim"])
    lr = float(g[p + "lr"])
    opt = argparse.Namespace(distill="moma", head="mlp", feat_dim=d, attn="self", mem="MoCo", nce_k=K_BIG, nce_t=0.15,
                             alpha=0.999, cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=0, multiprocessing_distributed=False,
                             print_freq=1000, batch_size=8, rank=0, world_size=1, s_dim=64, t_dim=64, moma_prec="bf16",
                             queue_dtype=queue_dtype, moma_fused=True, trace=[], overlap_teacher=overlap,
                             graph_teacher=overlap)
    assert _lib.load().moma_infonce_fused_workspace_bytes(8, d, K_BIG, 1, 1) > 0
    dev = torch.device("cuda", 0)
    ms, mt = resnet8(num_classes=100), resnet8(num_classes=100)
    ms.load_state_dict(sd(g, p + "s.")); mt.load_state_dict(sd(g, p + "t."))
    contrast = build_mem(opt)
    contrast.memory.copy_(big_queue(g, p, d).to(contrast.memory.dtype))
    kd = CMO(opt)
    fill_attention_(kd, g, p)
    ms, mt, contrast, kd = ms.to(dev), mt.to(dev), contrast.to(dev), kd.to(dev)
    trainer = ContrastTrainer(opt)
    trainable = nn.ModuleList([ms, kd.atts_q, kd.atts_k, kd.atts_queue, kd.embed_s])
    optimizer = torch.optim.SGD(trainable.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    mods = nn.ModuleList([ms, mt])
    crits = nn.ModuleList([nn.CrossEntropyLoss(), DistillKL(opt.kd_T), kd])
    images, labels = batches(g, p)
    torch.manual_seed(int(g[p + "loop_seed"]))
    for ep in range(2):
        loader = [(images[ep * 5 + i], labels[ep * 5 + i]) for i in range(5)]
        train_distill_moma(ep + 1, loader, mods, crits, trainer, contrast, optimizer, opt)
    losses = np.array([float(t[0]) for t in opt.trace])
    kds = np.array([float(t[2]) for t in opt.trace])
    assert [t[1] for t in opt.trace] == [int(v) for v in g[p + "index"]]
    ref_kd, ref_loss = g[p + "loss_kd"], g[p + "loss"]
    print("loss_kd |err|:", np.abs(kds - ref_kd).round(5), " total |err|:", np.abs(losses - ref_loss).round(5))
    # first step: no weight update in between -> the kernels' own error (north star: loss within 1e-3 of the reference)
    assert abs(kds[0] - ref_kd[0]) < 1e-3 * abs(ref_kd[0]) and abs(losses[0] - ref_loss[0]) < 1e-3 * abs(ref_loss[0])
    # later steps compound bf16 gradient rounding through the SGD updates; at lr = 0.05 loss_kd collapses 10.8 -> 1.06 -> 1e-4
    # within three steps and the trajectory amplifies it (observed up to 2.8e-3 at the 1.06 step), at lr = 0.002 it stays < 1e-3
    tol = 1e-3 if lr < 0.01 else 5e-3
    np.testing.assert_allclose(kds, ref_kd, rtol=tol, atol=tol)
    np.testing.assert_allclose(losses, ref_loss, rtol=tol, atol=5e-3)
    rows = contrast.memory[:80].float().cpu().numpy()
    ref_rows = g[p + "memory_rows_final"]
    # enqueued keys went through bf16 arithmetic (attention module) and, with a bf16 queue, bf16 storage
    np.testing.assert_allclose(rows, ref_rows, rtol=0, atol=3e-2 * np.abs(ref_rows).max())
    if contrast.memory.dtype == torch.float32:            # the mirror holds exactly the bf16 rounding of the fp32 queue
        assert torch.equal(contrast._shadow[:80], contrast.memory[:80].to(torch.bfloat16))
        assert torch.equal(contrast._shadow[80:4096], contrast.memory[80:4096].to(torch.bfloat16))
    assert all(q.grad is None for q in kd.atts_k.parameters()) and all(q.grad is None for q in kd.atts_queue.parameters())
    np.testing.assert_allclose(kd.atts_q.proj.weight.detach()[:8, :8].cpu().numpy(), g[p + "kd_final.atts_q.proj.weight_8x8"],
                               rtol=0, atol=2e-3)


@pytest.mark.parametrize("prec,queue_dtype,overlap", [("bf16", "bf16", True), ("bf16", "fp32", True), ("bf16", "bf16", False),
                                                      ("fp32", "fp32", True)])
def test_loop_bf16_policy_matches_reference_trace_bench_batch(golden_dir, prec, queue_dtype, overlap):
    """The benchmark's KD configuration at the benchmark's BATCH, tied to the reference at loop level (G5c: B = 256 -- two
    128-row blocks in the one-pass K2, eight key tiles per K1 workgroup --, K = 65536, --head mlp, d = 512, lr 0.002, 5 steps):
    bf16 policy (one-pass K2 fed with the query packed by atts_q's proj epilogue, K1 fast path with atts_k + atts_queue as one
    group, K3, K4), teacher side on the second stream + HIP graphs when overlap is on.  resnet8 backbones stay fp32.
    North-star tolerance on EVERY step: per-step loss_kd within 1e-3 relative of the reference; pointer exact.
    prec = fp32: the reference's own arithmetic -- K1 staged on f32-input MFMA, K2 as ONE pass over the fp32 queue on the f32 MFMA
    (infonce_f32.hip) -- at 1e-4."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tests.g5b_util import K_BIG, batches, big_queue, fill_attention_, sd
    from moma_amd.backbones.resnet_cifar import resnet8
    from moma_amd.MoMA.mem_moco import build_mem
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.distiller_zoo import DistillKL

    torch.backends.cudnn.benchmark = False
    g = np.load(os.path.join(golden_dir, "g5c_step_trace_b256.npz"))
    p = "c0_"
    d, B, steps, lr = int(g[p + "feat_dim"]), int(g[p + "B"]), int(g[p + "steps"]), float(g[p + "lr"])
    opt = argparse.Namespace(distill="moma", head="mlp", feat_dim=d, attn="self", mem="MoCo", nce_k=K_BIG, nce_t=0.15,
                             alpha=0.999, cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=0, multiprocessing_distributed=False,
                             print_freq=1000, batch_size=B, rank=0, world_size=1, s_dim=64, t_dim=64, moma_prec=prec,
                             queue_dtype=queue_dtype, moma_fused=True, trace=[], overlap_teacher=overlap,
                             graph_teacher=overlap)
    dev = torch.device("cuda", 0)
    ms, mt = resnet8(num_classes=100), resnet8(num_classes=100)
    ms.load_state_dict(sd(g, p + "s.")); mt.load_state_dict(sd(g, p + "t."))
    contrast = build_mem(opt)
    contrast.memory.copy_(big_queue(g, p, d).to(contrast.memory.dtype))
    kd = CMO(opt)
    fill_attention_(kd, g, p)
    ms, mt, contrast, kd = ms.to(dev), mt.to(dev), contrast.to(dev), kd.to(dev)
    trainer = ContrastTrainer(opt)
    trainable = nn.ModuleList([ms, kd.atts_q, kd.atts_k, kd.atts_queue, kd.embed_s])
    optimizer = torch.optim.SGD(trainable.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    mods = nn.ModuleList([ms, mt])
    crits = nn.ModuleList([nn.CrossEntropyLoss(), DistillKL(opt.kd_T), kd])
    images, labels = batches(g, p, steps, B)
    torch.manual_seed(int(g[p + "loop_seed"]))
    train_distill_moma(1, [(images[i], labels[i]) for i in range(steps)], mods, crits, trainer, contrast, optimizer, opt)
    losses = np.array([float(t[0]) for t in opt.trace])
    kds = np.array([float(t[2]) for t in opt.trace])
    assert [t[1] for t in opt.trace] == [int(v) for v in g[p + "index"]]
    ref_kd, ref_loss = g[p + "loss_kd"], g[p + "loss"]
    print("loss_kd rel err:", (np.abs(kds - ref_kd) / np.abs(ref_kd)).round(6), " total |err|:", np.abs(losses - ref_loss).round(5))
    tol = 1e-3 if prec == "bf16" else 1e-4
    np.testing.assert_allclose(kds, ref_kd, rtol=tol, atol=0)            # north star (bf16), every step
    np.testing.assert_allclose(losses, ref_loss, rtol=tol, atol=0)
    ids = torch.from_numpy(g[p + "memory_rows_ids"]).to(dev)
    rows = contrast.memory[ids].float().cpu().numpy()
    ref_rows = g[p + "memory_rows_final"]
    np.testing.assert_allclose(rows, ref_rows, rtol=0, atol=(3e-2 if prec == "bf16" else 1e-4) * np.abs(ref_rows).max())
    np.testing.assert_allclose(kd.atts_q.proj.weight.detach()[:8, :8].cpu().numpy(), g[p + "kd_final.atts_q.proj.weight_8x8"],
                               rtol=0, atol=2e-3)


def test_gather_mode_matches_reference_at_world_size_2_on_gpu(golden_dir, tmp_path):
    """n3 on the hardware path: two processes on ONE GPU (gloo carries the collectives; RCCL refuses two ranks on one device),
    `--shuffle_bn gather` with the HIP kernels underneath (K1 staged fp32 path, K2 logits, K3 enqueue), against the vectors the
    REFERENCE produced on two ranks (G9): pointer exact, enqueue order exact (same rows in the same slots on both ranks),
    keys / logits within fp32 kernel tolerance.  The CPU twin (tests/test_dp_gloo.py) checks the host logic alone."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import socket
    import torch.multiprocessing as mp
    from tests._g9_worker import compare, run_rank
    golden = os.path.join(golden_dir, "g9_gather_w2.npz")
    out = str(tmp_path / "g9")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(run_rank, args=(2, port, golden, "cuda", out), nprocs=2, join=True)
    compare(golden, out, 2, atol_k=2e-4, atol_logits=2e-3)


def test_shuffle_bn_gather_mode_single_rank_equals_per_rank(golden_dir):
    """`--shuffle_bn gather` (the reference's collectives C3-C5: image all_gather, id broadcast, key all_gather; reference
    learning/contrast_trainer.py:90-133) on a one-rank RCCL process group must reproduce the per-rank mode bit for bit:
    same permutation stream, same keys, same queue.  Runs the real collectives on the GPU (backend 'nccl' = RCCL)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import socket
    import torch.distributed as dist
    from moma_amd.backbones.resnet_cifar import resnet8
    from moma_amd.MoMA.mem_moco import build_mem
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.distiller_zoo import DistillKL

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    own_pg = not dist.is_initialized()
    if own_pg:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    try:
        dev = torch.device("cuda", 0)
        torch.backends.cudnn.benchmark = False
        results = {}
        for mode in ("per_rank", "gather"):
            opt = argparse.Namespace(distill="moma", head="mlp", feat_dim=128, attn="self", mem="MoCo", nce_k=4096, nce_t=0.15,
                                     alpha=0.999, cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=0, multiprocessing_distributed=True,
                                     print_freq=1000, batch_size=16, rank=0, local_rank=0, node_rank=0, ngpus_per_node=1,
                                     world_size=1, s_dim=64, t_dim=64, moma_prec="bf16", queue_dtype="bf16", moma_fused=True,
                                     trace=[], overlap_teacher=False, graph_teacher=False, shuffle_bn=mode)
            torch.manual_seed(0)
            ms, mt = resnet8(num_classes=10).to(dev), resnet8(num_classes=10).to(dev)
            contrast = build_mem(opt).to(dev)
            kd = CMO(opt).to(dev)
            trainer = ContrastTrainer(opt)
            trainer.local_group = dist.new_group([0])
            trainable = nn.ModuleList([ms, kd.atts_q, kd.atts_k, kd.atts_queue, kd.embed_s])
            optimizer = torch.optim.SGD(trainable.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
            crits = nn.ModuleList([nn.CrossEntropyLoss(), DistillKL(4.0), kd])
            g = torch.Generator().manual_seed(5)
            loader = [(torch.randn(16, 3, 32, 32, generator=g), torch.randint(0, 10, (16,), generator=g)) for _ in range(4)]
            torch.manual_seed(77)
            train_distill_moma(1, loader, nn.ModuleList([ms, mt]), crits, trainer, contrast, optimizer, opt)
            results[mode] = (torch.stack([t[0] for t in opt.trace]).cpu(), contrast.memory.clone().cpu(), contrast.index,
                             kd.atts_q.proj.weight.detach().clone().cpu())
        a, b = results["per_rank"], results["gather"]
        assert a[2] == b[2] == 64
        assert torch.equal(a[1], b[1])                  # same keys enqueued in the same (shuffled) order
        # (the student's convolution weight gradients come from MIOpen kernels that are not bitwise reproducible run to run,
        #  so from the second step on the two runs differ in the last bits; the first step and the keys are exact)
        assert a[0][0] == b[0][0]
        assert torch.allclose(a[0], b[0], rtol=0, atol=1e-3) and torch.allclose(a[3], b[3], rtol=0, atol=1e-4)
    finally:
        if own_pg:
            dist.destroy_process_group()


@pytest.mark.parametrize("ci", [0, 1, 2])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_shuffle_bn_attn_matches_reference(golden_dir, ci, prec):
    """ContrastTrainer._shuffle_bn_attn (attention over [q ; k] / per side before the un-shuffle) against the vectors captured
    from the reference (G8; learning/contrast_trainer.py:135-187): outputs, dq, and the gradients of the attention weights."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.backbones.resnet_cifar import resnet8
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    torch.backends.cudnn.benchmark = False
    g = np.load(os.path.join(golden_dir, "g8_shuffle_bn_attn.npz"))
    p = f"c{ci}_"
    attn, head, d = str(g[p + "attn"]), str(g[p + "head"]), int(g[p + "d"])
    opt = argparse.Namespace(head=head, s_dim=64, t_dim=64, feat_dim=d, attn=attn, local_rank=0, node_rank=0, ngpus_per_node=1,
                             rank=0, world_size=1, moma_prec=prec, shuffle_bn="per_rank")
    dev = torch.device("cuda", 0)
    mt = resnet8(num_classes=10)
    mt.load_state_dict(_sd(g, p + "t."))
    kd = CMO(opt)
    kd.load_state_dict(_sd(g, p + "kd."))
    mt, kd = mt.to(dev).train(), kd.to(dev)
    trainer = ContrastTrainer(opt)
    q = torch.from_numpy(g[p + "q"]).to(dev).requires_grad_(True)
    torch.manual_seed(int(g[p + "perm_seed"]))
    q2, k, all_k = trainer._shuffle_bn_attn(torch.from_numpy(g[p + "x"]).to(dev), mt, kd.embed_t, kd, q)
    w = [torch.from_numpy(g[p + n]).to(dev) for n in ("w1", "w2", "w3")]
    ((q2 * w[0]).sum() + (k * w[1]).sum() + (all_k * w[2]).sum()).backward()
    rt = 3e-5 if prec == "fp32" else 2e-2

    def close(a, ref):
        np.testing.assert_allclose(a.detach().cpu().numpy(), ref, rtol=0, atol=rt * max(1e-6, np.abs(ref).max()))
    close(q2, g[p + "q_out"]); close(k, g[p + "k_out"]); close(all_k, g[p + "all_k"])
    close(q.grad, g[p + "dq"])
    for name, prm in kd.named_parameters():
        key = p + "grad." + name
        if key in g.files:
            close(prm.grad, g[key])
        else:
            assert prm.grad is None, name


@pytest.mark.parametrize("mem,attn", [("MoCoAtt", "qk"), ("MoCoAtt", "all"), ("MoCoAtt", "self_qk"), ("MoCoAtt", "self"),
                                      ("MoCoAtt", "dual"), ("MoCo", "self_mix"), ("MoCo", "self_nomix")])
def test_loop_cross_attention_paths_match_step_oracle(mem, attn):
    """The loop paths the reference CLI cannot reach: `--mem MoCoAtt` (the memory applies the
    cross-attention variant: MoMA/mem_moco.py:111-161) and `--attn self_mix|self_nomix` (Shuffle-BN with the attention before
    the un-shuffle: learning/contrast_trainer.py:135-187).  4 steps of train_distill_moma (fp32 policy) against the CPU step
    oracle, whose MoCoAtt / shuffle_bn_attn restatements are pinned to the reference by G7 / G8: per-step loss and loss_kd,
    pointer, final queue."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.step_oracle import OracleCMO, OracleMoCo, OracleMoCoAtt, StepOracle
    from moma_amd.backbones.resnet_cifar import resnet8
    from moma_amd.MoMA.mem_moco import build_mem
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.distiller_zoo import DistillKL
    torch.backends.cudnn.benchmark = False
    K, d, B, lr = 96, 32, 8, 0.01
    opt = argparse.Namespace(distill="moma", head="mlp", feat_dim=d, attn=attn, mem=mem, nce_k=K, nce_t=0.15, alpha=0.999,
                             cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=0, multiprocessing_distributed=False, print_freq=1000,
                             batch_size=B, rank=0, world_size=1, s_dim=64, t_dim=64, moma_prec="fp32", moma_fused=True, trace=[],
                             overlap_teacher=True, graph_teacher=False, local_rank=0, node_rank=0, ngpus_per_node=1)
    torch.manual_seed(321)
    ms, mt = resnet8(num_classes=10), resnet8(num_classes=10)
    contrast = build_mem(opt)
    kd = CMO(opt)
    # the oracle twin, same weights
    import copy
    oms, omt = copy.deepcopy(ms), copy.deepcopy(mt)
    ocmo = OracleCMO("mlp", 64, 64, d, attn=attn)
    ocmo.load_state_dict(kd.state_dict())
    ocontrast = (OracleMoCoAtt if mem == "MoCoAtt" else OracleMoCo)(d, K, 0.15)
    ocontrast.memory.copy_(contrast.memory)
    run = StepOracle(oms, omt, ocmo, ocontrast, head="mlp", lr=lr, attn=attn)
    dev = torch.device("cuda", 0)
    ms, mt, contrast, kd = ms.to(dev), mt.to(dev), contrast.to(dev), kd.to(dev)
    trainer = ContrastTrainer(opt)
    names = [n for n in ("atts", "atts_p", "atts_n", "atts_q", "atts_k", "atts_queue") if hasattr(kd, n)]
    trainable = nn.ModuleList([ms] + [getattr(kd, n) for n in names] + [kd.embed_s])
    optimizer = torch.optim.SGD(trainable.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    crits = nn.ModuleList([nn.CrossEntropyLoss(), DistillKL(4.0), kd])
    g = torch.Generator().manual_seed(9)
    loader = [(torch.randn(B, 3, 32, 32, generator=g), torch.randint(0, 10, (B,), generator=g)) for _ in range(4)]
    torch.manual_seed(55)
    train_distill_moma(1, loader, nn.ModuleList([ms, mt]), crits, trainer, contrast, optimizer, opt)
    torch.manual_seed(55)
    run.start_epoch()
    ref = [run.step(x, y) for x, y in loader]
    losses = [float(t[0]) for t in opt.trace]
    kds = [float(t[2]) for t in opt.trace]
    assert [t[1] for t in opt.trace] == [(i + 1) * B % K for i in range(4)] and ocontrast.index == contrast.index
    np.testing.assert_allclose(losses, [r[0] for r in ref], rtol=0, atol=2e-3)        # (backbone: MIOpen vs CPU convolutions)
    np.testing.assert_allclose(kds, [r[2] for r in ref], rtol=0, atol=2e-3)
    np.testing.assert_allclose(contrast.memory.cpu().numpy(), ocontrast.memory.numpy(), rtol=0, atol=2e-3)


@pytest.mark.parametrize("d,K,prec", [(768, 8192, "bf16"), (1280, 16384, "bf16"), (2048, 8192, "bf16"),
                                      (1280, 16384, "fp32"), (384, 8192, "fp32")])
def test_loop_wide_queue_bf16_policy_matches_step_oracle(d, K, prec):
    """Wide feature dims (the regime of the reference CLI's default `--head None`: EfficientNet-B0 -> d = 1280, ResNet-50 -> 2048) at
    LOOP level under the bf16 policy: the two-pass wide-row K2 (infonce_wide_scores_kernel -- at d = 2048 in two register passes of
    Q -- + infonce_wide_pv2_kernel, bf16 queue), K1 on the wide-head fast path (head dim 192 / 320 / 512 > 128: segment-streamed
    cores), K3, K4; teacher side on the second stream.  3 steps of
    train_distill_moma against the CPU step oracle (fp32; its pieces are pinned to the reference by G1-G5).  Tolerances as in
    the big-queue test: first step 1e-3 relative (the kernels' own error), later steps 5e-3 (bf16 gradient rounding through SGD).
    prec = fp32 (round 4): the same loop in the reference's own arithmetic at the reference CLI's default width -- the exact-fp32
    one-pass K2 over the fp32 queue (infonce_f32.hip, 5 / 3 column segments per wave at d = 1280 / 384: no [B,K+1] logits), staged
    exact-fp32 K1 -- loss_kd within 2e-5 relative on every step."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import copy
    from oracle.step_oracle import OracleCMO, OracleMoCo, StepOracle
    from moma_amd.backbones.resnet_cifar import resnet8
    from moma_amd.MoMA.mem_moco import build_mem
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.distiller_zoo import DistillKL
    torch.backends.cudnn.benchmark = False
    B, lr = 8, 0.002
    opt = argparse.Namespace(distill="moma", head="mlp", feat_dim=d, attn="self", mem="MoCo", nce_k=K, nce_t=0.15, alpha=0.999,
                             cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=0, multiprocessing_distributed=False, print_freq=1000,
                             batch_size=B, rank=0, world_size=1, s_dim=64, t_dim=64, moma_prec=prec,
                             queue_dtype="bf16" if prec == "bf16" else "fp32",
                             moma_fused=True, trace=[], overlap_teacher=True, graph_teacher=False, local_rank=0, node_rank=0,
                             ngpus_per_node=1)
    if prec == "fp32":
        from moma_amd import _lib
        # the one-pass kernel takes the shape: its workspace is the chunk partials, not the staged path's [B,K+1] logits matrix
        assert _lib.load().moma_infonce_fused_workspace_bytes(B, d, K, 0, 0) != (B * (K + 1) * 4 + 255) // 256 * 256
    torch.manual_seed(4321 + d)
    ms, mt = resnet8(num_classes=10), resnet8(num_classes=10)
    contrast = build_mem(opt)
    kd = CMO(opt)
    oms, omt = copy.deepcopy(ms), copy.deepcopy(mt)
    ocmo = OracleCMO("mlp", 64, 64, d, attn="self")
    ocmo.load_state_dict(kd.state_dict())
    ocontrast = OracleMoCo(d, K, 0.15)
    ocontrast.memory.copy_(contrast.memory.float())              # (the oracle sees the bf16-rounded queue the kernels read)
    run = StepOracle(oms, omt, ocmo, ocontrast, head="mlp", lr=lr, attn="self")
    dev = torch.device("cuda", 0)
    ms, mt, contrast, kd = ms.to(dev), mt.to(dev), contrast.to(dev), kd.to(dev)
    trainer = ContrastTrainer(opt)
    trainable = nn.ModuleList([ms, kd.atts_q, kd.atts_k, kd.atts_queue, kd.embed_s])
    optimizer = torch.optim.SGD(trainable.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4)
    crits = nn.ModuleList([nn.CrossEntropyLoss(), DistillKL(4.0), kd])
    g = torch.Generator().manual_seed(19)
    loader = [(torch.randn(B, 3, 32, 32, generator=g), torch.randint(0, 10, (B,), generator=g)) for _ in range(3)]
    torch.manual_seed(77)
    train_distill_moma(1, loader, nn.ModuleList([ms, mt]), crits, trainer, contrast, optimizer, opt)
    torch.manual_seed(77)
    run.start_epoch()
    ref = [run.step(x, y) for x, y in loader]
    kds = np.array([float(t[2]) for t in opt.trace])
    losses = np.array([float(t[0]) for t in opt.trace])
    ref_kd, ref_loss = np.array([r[2] for r in ref]), np.array([r[0] for r in ref])
    print("loss_kd |err|:", np.abs(kds - ref_kd).round(5), " total |err|:", np.abs(losses - ref_loss).round(5))
    assert [t[1] for t in opt.trace] == [(i + 1) * B % K for i in range(3)] and ocontrast.index == contrast.index
    if prec == "fp32":
        np.testing.assert_allclose(kds, ref_kd, rtol=2e-5, atol=0)
        np.testing.assert_allclose(losses, ref_loss, rtol=5e-5, atol=0)
    else:
        assert abs(kds[0] - ref_kd[0]) < 1e-3 * abs(ref_kd[0])
        np.testing.assert_allclose(kds, ref_kd, rtol=5e-3, atol=5e-3)
        np.testing.assert_allclose(losses, ref_loss, rtol=5e-3, atol=5e-3)
    rows = contrast.memory[:3 * B].float().cpu().numpy()
    ref_rows = ocontrast.memory[:3 * B].numpy()
    np.testing.assert_allclose(rows, ref_rows, rtol=0, atol=(3e-2 if prec == "bf16" else 1e-4) * np.abs(ref_rows).max())


def test_mlp_byol_head_runs_the_kd_term_on_the_gpu():
    """`--head mlp_byol` (reference MoMA/criterion_moco_att.py:269-283: Linear - BatchNorm1d - ReLU - Linear - L2; the class offers
    it, the reference CLI's choices do not): heads -> atts_q -> one-pass K2 (+ query packed by K1) -> enqueue, backward into the
    head's BatchNorm1d and Linear weights, against the same chain in plain torch fp32."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.MoMA.mem_moco import MoCo
    torch.manual_seed(3)
    B, s_dim, d, K, T = 64, 96, 128, 4096, 0.15
    opt = argparse.Namespace(head="mlp_byol", s_dim=s_dim, t_dim=s_dim, feat_dim=d, attn="self", moma_prec="bf16", num_heads=4)
    kd = CMO(opt).cuda().train()
    assert any(isinstance(m, nn.BatchNorm1d) for m in kd.embed_s.modules())
    mem = MoCo(d, K, T, queue_dtype=torch.bfloat16, precision="bf16").cuda()
    queue0 = mem.memory.float().clone()
    feat_s = torch.randn(B, s_dim, 1, 1, device="cuda")
    feat_t = torch.randn(B, s_dim, 1, 1, device="cuda")
    with torch.no_grad():
        k = kd.atts_k(kd.embed_t(feat_t))
    qp = mem.qpack(B, d, feat_s.device)
    q = kd.atts_q(kd.embed_s(feat_s), qpack=qp)
    loss, _ = mem.forward_fused(q, k, qpack=qp)
    loss.backward()
    # plain torch fp32 of the same chain (reference op order)
    ref = CMO(argparse.Namespace(**{**vars(opt), "moma_prec": "fp32"})).cuda().train()
    ref.load_state_dict(kd.state_dict())

    def att(m, x):
        n, c = x.shape
        h = m.num_heads
        qkv = nn.functional.linear(x, m.qkv.weight, m.qkv.bias).reshape(n, 3, h, c // h).permute(1, 2, 0, 3)
        a = ((qkv[0] @ qkv[1].transpose(-2, -1)) * m.scale).softmax(dim=-1)
        return nn.functional.linear((a @ qkv[2]).transpose(0, 1).reshape(n, c), m.proj.weight, m.proj.bias)
    rq = att(ref.atts_q, ref.embed_s(feat_s))
    logits = torch.cat([(rq * k).sum(1, keepdim=True), rq @ queue0.t()], dim=1) / T
    rloss = nn.functional.cross_entropy(logits, torch.zeros(B, dtype=torch.long, device="cuda"))
    rloss.backward()
    assert abs(loss.item() - rloss.item()) < 1e-3 * abs(rloss.item()), (loss.item(), rloss.item())
    gmax = max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)
    for (n0, p0), (n1, p1) in zip(kd.named_parameters(), ref.named_parameters()):
        if n0.startswith("embed_s.") or n0.startswith("atts_q."):
            assert p0.grad is not None and p1.grad is not None, n0
            # relative to the tensor's own scale, with a floor: the bias in front of the BatchNorm1d has a gradient that is zero
            # up to rounding (the normalisation removes it)
            err = (p0.grad - p1.grad).abs().max().item() / max(p1.grad.abs().max().item(), 1e-2 * gmax)
            assert err < 5e-2, (n0, err)
    assert mem.index == B
