"""Data-parallel host logic on CPU: 2 processes, gloo, 127.0.0.1.

What is under test is the HOST orchestration of the N>1 path (BASELINE configs[3]): DDP on the student, the
explicit flat all-reduce of the trainable criterion modules (fix of SURVEY Q7), per-rank queue + pointer, EMA
teacher staying identical across ranks without communication, metric reduction, and the reference-faithful
`gather` Shuffle-BN mode (collectives C3-C5).  There is no GPU here, and the product has no CPU path, so inside
the spawned workers the four kernel entry points of `moma_amd.ops` are replaced by the torch-CPU restatement
(same op order as `oracle/step_oracle.py`) -- a test-only stand-in; the product never routes through it."""
import argparse
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _install_cpu_standins():
    """Replace the C-ABI wrappers by torch-CPU restatements (reference op order) for host-logic tests."""
    from moma_amd import ops

    def mha(x, w_qkv, b_qkv, w_proj, b_proj, num_heads, prec="fp32", pack=None, qpack=None):
        n, c = x.shape
        qkv = F.linear(x, w_qkv, b_qkv).reshape(n, 3, num_heads, c // num_heads).permute(1, 2, 0, 3)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = ((q @ k.transpose(-2, -1)) * (c // num_heads) ** -0.5).softmax(dim=-1)
        return F.linear((a @ v).transpose(0, 1).reshape(n, c), w_proj, b_proj)

    def mha_group(calls, num_heads, prec="fp32"):
        return [mha(*c[:5], num_heads, prec) for c in calls]

    def infonce_logits(q, k, queue, T, prec="fp32"):
        return torch.cat([(q * k).sum(1, keepdim=True), q @ queue.float().t()], dim=1) / T

    def enqueue_(queue, rows, index):
        K = queue.shape[0]
        ids = torch.fmod(torch.arange(rows.shape[0]) + index, K).long()
        queue.index_copy_(0, ids, rows.to(queue.dtype))

    def infonce_fused(q, k, queue, T, prec="fp32", qpack=None, enq=None):
        logits = torch.cat([(q * k).sum(1, keepdim=True), q @ queue.float().clone().t()], dim=1) / T   # pre-enqueue snapshot
        lse = torch.logsumexp(logits, dim=1)
        top1 = (logits[:, 0] >= logits.max(dim=1).values).to(torch.int32)
        if enq is not None:                                   # the enqueue the K2 call carries: read old, then enqueue
            rows, index, full = enq
            with torch.no_grad():
                enqueue_(queue, rows, index)
                if full is not None:
                    enqueue_(full, rows, index)
        return lse - logits[:, 0], lse.detach(), top1

    class EmaTable:
        def __init__(self, ps, es):
            self.ps, self.es = list(ps), list(es)

        def matches(self, ps, es):
            return True

    def ema_update_(table, m):
        with torch.no_grad():
            for p, e in zip(table.ps, table.es):
                e.mul_(m).add_(p, alpha=1 - m)

    ops.mha, ops.infonce_fused, ops.enqueue_, ops.EmaTable, ops.ema_update_ = mha, infonce_fused, enqueue_, EmaTable, ema_update_
    ops.mha_group, ops.infonce_logits = mha_group, infonce_logits


def _worker(rank, world, port, shuffle_mode, out, dp="flat"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_cpu_standins()
    from moma_amd.backbones.resnet_cifar import resnet8
    from moma_amd.MoMA.mem_moco import build_mem
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.helper.util import reduce_tensor
    from moma_amd.distiller_zoo import DistillKL

    B, K = 4, 24
    opt = argparse.Namespace(distill="moma", head="mlp", feat_dim=32, attn="self", mem="MoCo", nce_k=K, nce_t=0.15,
                             alpha=0.9, cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=None, device=torch.device("cpu"),
                             multiprocessing_distributed=True, print_freq=10 ** 9, batch_size=B, rank=rank,
                             local_rank=rank, node_rank=0, ngpus_per_node=world, world_size=world, s_dim=64, t_dim=64,
                             moma_prec="fp32", moma_fused=True, shuffle_bn=shuffle_mode, trace=[])
    torch.manual_seed(0)                                   # identical initial weights on every rank
    ms, mt = resnet8(num_classes=10), resnet8(num_classes=10)
    contrast = build_mem(opt)
    kd = CMO(opt)
    trainer = ContrastTrainer(opt)
    trainer.local_group = dist.new_group(list(range(world)))
    trainer.broadcast_memory(contrast)
    trainable = nn.ModuleList([ms, kd.atts_q, kd.atts_k, kd.atts_queue, kd.embed_s])
    optimizer = torch.optim.SGD(trainable.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    from moma_amd.learning.ddp import wrap_student
    ddp = wrap_student(ms, mode=dp)                         # the product's wrap (default: FlatDataParallel; 'ddp' = stock reducer)
    mods = [ddp, mt]
    crits = nn.ModuleList([nn.CrossEntropyLoss(), DistillKL(4.0), kd])
    g = torch.Generator().manual_seed(100 + rank)          # different data shard per rank
    loader = [(torch.randn(B, 3, 16, 16, generator=g), torch.randint(0, 10, (B,), generator=g)) for _ in range(3)]
    torch.manual_seed(7 + (0 if shuffle_mode == "gather" else rank))
    acc, loss = train_distill_moma(1, loader, mods, crits, trainer, contrast, optimizer, opt)
    red = reduce_tensor(torch.tensor([acc, loss]), world).tolist()

    def flat(mod):
        return torch.cat([p.detach().reshape(-1) for p in mod.parameters()])

    res = dict(index=contrast.index, student=flat(ms), teacher=flat(mt), atts_q=flat(kd.atts_q), embed_s=flat(kd.embed_s),
               atts_k=flat(kd.atts_k), memory=contrast.memory.clone(), red=red, loss=loss,
               atts_k_grad_none=all(p.grad is None for p in kd.atts_k.parameters()))
    if shuffle_mode == "gather":                             # one more call to inspect the gathered keys
        x = loader[0][0]
        k, all_k = trainer._shuffle_bn(x, mt, kd.embed_t)
        res.update(k_shape=tuple(k.shape), all_k_shape=tuple(all_k.shape))
    torch.save(res, os.path.join(out, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shuffle_mode,dp", [("per_rank", "flat"), ("gather", "flat"), ("per_rank", "ddp")])
def test_two_rank_data_parallel(tmp_path, shuffle_mode, dp):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, shuffle_mode, str(tmp_path), dp), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    B, K = 4, 24
    # replicas stay in sync: student through DDP, criterion modules through the explicit flat all-reduce
    assert torch.equal(r0["student"], r1["student"])
    assert torch.allclose(r0["atts_q"], r1["atts_q"], atol=1e-7) and torch.allclose(r0["embed_s"], r1["embed_s"], atol=1e-7)
    # EMA teacher is a function of the (identical) student only -> identical without communication
    assert torch.allclose(r0["teacher"], r1["teacher"], atol=1e-7)
    # atts_k never gets a gradient -> untouched and equal
    assert r0["atts_k_grad_none"] and r1["atts_k_grad_none"] and torch.equal(r0["atts_k"], r1["atts_k"])
    n_enq = B if shuffle_mode == "per_rank" else B * world
    assert r0["index"] == r1["index"] == (3 * n_enq) % K
    if shuffle_mode == "per_rank":
        assert not torch.equal(r0["memory"], r1["memory"])           # per-rank queues hold per-rank keys
    else:
        assert torch.allclose(r0["memory"], r1["memory"], atol=1e-6)  # reference mode: every rank enqueues all keys
        assert r0["all_k_shape"] == (B * world, 32) and r0["k_shape"] == (B, 32)
    # epoch metrics are all-reduced (avg)
    assert r0["red"] == r1["red"] and abs(r0["red"][1] - 0.5 * (r0["loss"] + r1["loss"])) < 1e-5


@pytest.mark.parametrize("shuffle_mode", ["per_rank", "gather"])
def test_four_rank_data_parallel(tmp_path, shuffle_mode):
    """The same at world size 4 (the driver's N = 4 step of the scaling run): replicas identical on every rank, per-rank queues
    pairwise different / gathered queues equal, pointer arithmetic with the global batch in gather mode, metrics averaged over 4."""
    world, port = 4, _free_port()
    mp.spawn(_worker, args=(world, port, shuffle_mode, str(tmp_path), "flat"), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"r{i}.pt") for i in range(world)]
    B, K = 4, 24
    for r in rs[1:]:
        assert torch.equal(rs[0]["student"], r["student"])
        assert torch.allclose(rs[0]["atts_q"], r["atts_q"], atol=1e-7) and torch.allclose(rs[0]["embed_s"], r["embed_s"], atol=1e-7)
        assert torch.allclose(rs[0]["teacher"], r["teacher"], atol=1e-7)
        assert r["red"] == rs[0]["red"]
    n_enq = B if shuffle_mode == "per_rank" else B * world
    assert all(r["index"] == (3 * n_enq) % K for r in rs)
    if shuffle_mode == "per_rank":
        assert all(not torch.equal(rs[i]["memory"], rs[j]["memory"]) for i in range(world) for j in range(i))
    else:
        assert all(torch.allclose(rs[0]["memory"], r["memory"], atol=1e-6) for r in rs[1:])
        assert rs[0]["all_k_shape"] == (B * world, 32)
    assert abs(rs[0]["red"][1] - sum(r["loss"] for r in rs) / world) < 1e-5


def test_gather_mode_matches_reference_at_world_size_2(tmp_path, golden_dir):
    """n3 pinned to the REFERENCE at W = 2 (G9: learning/contrast_trainer.py:90-187 + MoMA/mem_moco.py:77-100 captured on two gloo
    ranks): `--shuffle_bn gather` -- image all_gather, id broadcast, teacher on the shuffled local share, key all_gather,
    un-shuffle, GLOBAL enqueue -- reproduces per-rank k / all_k / logits / queue / pointer for _shuffle_bn and for
    _shuffle_bn_attn (self_nomix, self_mix).  Host logic on CPU (torch stand-ins for the kernels); the GPU twin of this test
    runs the HIP kernels underneath."""
    from tests._g9_worker import compare, run_rank
    golden = os.path.join(golden_dir, "g9_gather_w2.npz")
    out = str(tmp_path / "g9")
    mp.spawn(run_rank, args=(2, _free_port(), golden, "cpu", out), nprocs=2, join=True)
    compare(golden, out, 2, atol_k=2e-5, atol_logits=2e-4)


def _grad_sync_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    torch.manual_seed(0)
    layers = nn.ModuleList([nn.Linear(4, 4) for _ in range(4)])          # the 4th never takes part (like atts_k: grad stays None)
    params = list(layers.parameters())
    trainer = ContrastTrainer(argparse.Namespace(rank=rank, local_rank=rank, world_size=world))
    trainer.attach_grad_sync(params)
    g = torch.Generator().manual_seed(10 + rank)
    ok = True
    # which layers take part per step: the count of gradient-receiving parameters goes 4 -> 6 (MORE than the hooks learnt) ->
    # 2 (fewer) -> 6 -> 6 (steady: launched from the hooks)
    for step, used in enumerate([(0, 1), (0, 1, 2), (0,), (0, 1, 2), (0, 1, 2)]):
        for p in params:
            p.grad = None
        x = torch.randn(3, 4, generator=g)
        sum(layers[i](x).pow(2).sum() for i in used).backward()
        local = [None if p.grad is None else p.grad.clone() for p in params]
        trainer.finish_grad_sync()
        for p, l in zip(params, local):
            if l is None:
                ok = ok and p.grad is None
                continue
            ref = l.clone()
            dist.all_reduce(ref)
            ok = ok and torch.allclose(p.grad, ref / world, rtol=0, atol=1e-7)
    torch.save({"ok": ok, "expect": trainer._gs_expect}, f"{out}.rank{rank}")
    dist.destroy_process_group()


def test_criterion_grad_sync_survives_a_changing_gradient_set(tmp_path):
    """the hook-launched flat all-reduce of the criterion gradients learns how many gradients a backward
    produces on the first step; a later backward that produces MORE must not leave the late ones un-reduced (and fewer
    must not hang): every step's gradients equal the rank average, whatever the set."""
    out = str(tmp_path / "gs")
    mp.spawn(_grad_sync_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(f"{out}.rank{r}")
        assert res["ok"] and res["expect"] == 6


def _wrap_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    from moma_amd.learning.ddp import wrap_student

    def net():
        torch.manual_seed(3)
        return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1),
                             nn.BatchNorm2d(8), nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 5))
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    a, b, c = net(), net(), net()
    if rank == 1:                                                        # replicas must START from rank 0 (the wraps broadcast)
        with torch.no_grad():
            for m in (a, b, c):
                for p in m.parameters():
                    p.add_(0.123)
    stock = nn.parallel.DistributedDataParallel(a)                       # the reference's wrap (stock defaults)
    ours = wrap_student(b, mode="ddp")                                   # stock reducer + flat buffer broadcast
    flat = wrap_student(c, mode="flat")                                  # the default: ONE flat gradient all-reduce per step
    opts = [torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9) for m in (a, b, c)]
    g = torch.Generator().manual_seed(50 + rank)                         # a different shard per rank
    for _ in range(3):
        x, y = torch.randn(6, 3, 8, 8, generator=g), torch.randint(0, 5, (6,), generator=g)
        for m, o in zip((stock, ours, flat), opts):
            o.zero_grad(set_to_none=True)
            nn.functional.cross_entropy(m(x), y).backward()
            if m is flat:
                assert flat.allreduce_grads(flat.grad_params()) == 1
            o.step()
    stock.eval(), ours.eval(), flat.eval()
    with torch.no_grad():                                                # one more forward (eval: no local update behind the
        stock(x), ours(x), flat(x)                                       # broadcast): every wrap leaves rank 0's buffers everywhere
    res = dict(pa=[p.detach().clone() for p in a.parameters()], pb=[p.detach().clone() for p in b.parameters()],
               pc=[p.detach().clone() for p in c.parameters()],
               ba=[t.clone() for t in a.buffers()], bb=[t.clone() for t in b.buffers()], bc=[t.clone() for t in c.buffers()])
    torch.save(res, os.path.join(out, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_student_wrap_equals_stock_ddp(tmp_path):
    """learning/ddp.py:wrap_student -- mode 'flat' (the default: FlatDataParallel + ONE flat gradient all-reduce per step) and mode
    'ddp' (stock reducer + one flat buffer broadcast per dtype) -- against stock DistributedDataParallel, the reference's wrap
    (train_student_moma.py:345-349), on two gloo ranks with different shards and DIFFERENT initial weights on rank 1: parameters
    after three SGD steps and BatchNorm buffers (incl. the int64 batch counters) are the same, on both ranks."""
    world, port = 2, _free_port()
    mp.spawn(_wrap_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    for r in (r0, r1):
        for pa, pb, pc in zip(r["pa"], r["pb"], r["pc"]):
            assert torch.allclose(pa, pb, rtol=0, atol=1e-7)
            assert torch.allclose(pa, pc, rtol=0, atol=2e-7)             # (sum, then divide -- the reducer divides, then sums)
        for ba, bb, bc in zip(r["ba"], r["bb"], r["bc"]):
            assert torch.equal(ba, bb)
            assert torch.allclose(ba.float(), bc.float(), rtol=0, atol=2e-7)
    for key in ("bb", "bc"):
        for b0, b1 in zip(r0[key], r1[key]):
            assert torch.equal(b0, b1)                                   # rank 0's statistics everywhere
    for key in ("pb", "pc"):
        for p0, p1 in zip(r0[key], r1[key]):
            assert torch.equal(p0, p1)                                   # replicas identical


def _safety_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from moma_amd.learning.ddp import (FlatDataParallel, broadcast_module_state, collective_self_test, wrap_student)
    res = {}
    # (1) the wrap's two collectives pass their self-test on this communicator; mode None / 'auto' then picks the flat wrap
    res["self_test"] = collective_self_test(torch.device("cpu"))
    torch.manual_seed(rank)
    res["auto_is_flat"] = isinstance(wrap_student(nn.Linear(4, 3), mode="auto"), FlatDataParallel)
    # (2) replicas with different layouts are refused at construction, on every rank
    try:
        FlatDataParallel(nn.Linear(4, 3 + rank))
        res["layout_refused"] = False
    except RuntimeError as e:
        res["layout_refused"] = "differ across ranks" in str(e)
    # (3) modules under no wrap (criterion, teacher): rank 0's state everywhere
    torch.manual_seed(10 + rank)
    m = nn.Sequential(nn.Linear(5, 5), nn.BatchNorm1d(5))
    m[1].running_mean.fill_(float(rank + 1))
    broadcast_module_state([m, None])
    res["state"] = torch.cat([t.detach().reshape(-1).double() for t in list(m.parameters()) + list(m.buffers())])
    # (4) a gradient set that differs across the ranks raises on every rank instead of reducing buffers of different sizes
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(4, 4), nn.Linear(4, 2))
    flat = FlatDataParallel(net)
    extra = nn.Linear(4, 1)                      # a "criterion module": receives a gradient on rank 1 only
    x = torch.randn(3, 4)
    loss = flat(x).sum() + (extra(x).sum() if rank == 1 else 0.0)
    loss.backward()
    params = flat.grad_params() + list(extra.parameters())
    try:
        flat.allreduce_grads(params)
        res["gradset_refused"] = False
    except RuntimeError as e:
        res["gradset_refused"] = "gradient sets differ" in str(e)
    # (5) same set on both ranks: one collective, averaged
    for p in params:
        p.grad = None
    (flat(x * (rank + 1)).sum() + extra(x * (rank + 1)).sum()).backward()
    local = [p.grad.clone() for p in params]
    res["n_coll"] = flat.allreduce_grads(params)
    gathered = [[torch.empty_like(g) for _ in range(world)] for g in local]
    for g, dst in zip(local, gathered):
        dist.all_gather(dst, g)
    res["avg_ok"] = all(torch.allclose(p.grad, sum(dst) / world, atol=1e-6) for p, dst in zip(params, gathered))
    # (6) AFTER the ranks have agreed once, the set changes on ONE rank only (round 4 re-verified only when the LOCAL
    #     set changed, so rank 1 went into the exchange alone while rank 0 entered the all-reduce): refused on every rank, every time
    for p in params:
        p.grad = None
    (flat(x).sum() + (extra(x).sum() if rank == 0 else 0.0)).backward()      # rank 0 keeps the agreed set, rank 1 loses `extra`
    try:
        flat.allreduce_grads(params)
        res["late_change_refused"] = False
    except RuntimeError as e:
        res["late_change_refused"] = "gradient sets differ" in str(e)
    torch.save(res, f"{out}.rank{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_flat_wrap_safety_checks_over_a_separate_control_group(tmp_path, monkeypatch):
    """The same checks with the host-side agreement channel as a gloo group OF ITS OWN next to the data group (what a job on RCCL
    gets: `dist.new_group(backend="gloo")` beside the nccl communicator), forced here on a gloo job by MOMA_DP_CONTROL=new."""
    monkeypatch.setenv("MOMA_DP_CONTROL", "new")
    out = str(tmp_path / "safe2")
    mp.spawn(_safety_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in (torch.load(f"{out}.rank0"), torch.load(f"{out}.rank1")):
        assert r["gradset_refused"] and r["late_change_refused"] and r["n_coll"] == 1 and r["avg_ok"]


def test_flat_wrap_safety_checks(tmp_path):
    """ADVICE r3 (learning/ddp.py): what stock DDP verifies and the flat wrap has to verify itself -- identical replica layouts at
    construction, identical gradient sets before the flat all-reduce (a mismatch raises on EVERY rank instead of hanging or
    corrupting), a self-test of the two collectives on the live communicator, and rank 0's state for the modules under no wrap."""
    out = str(tmp_path / "safe")
    mp.spawn(_safety_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(f"{out}.rank0"), torch.load(f"{out}.rank1")
    for r in (r0, r1):
        assert r["self_test"] and r["auto_is_flat"] and r["layout_refused"] and r["gradset_refused"]
        assert r["n_coll"] == 1 and r["avg_ok"] and r["late_change_refused"]
    assert torch.equal(r0["state"], r1["state"]) and r0["state"][-11:-6].eq(1.0).all()     # rank 0's running_mean (rank 1 had 2.0); then var [5], count [1]


def _ctl_fail_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import moma_amd.learning.ddp as ddp
    res = {}
    real = ddp.control_group

    def flaky(group=None, force_new=False):
        if rank == 1:
            raise RuntimeError("no usable gloo transport on this rank")
        return real(group, force_new)
    ddp.control_group = flaky
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(4, 4), nn.Linear(4, 2))
    flat = ddp.FlatDataParallel(net)
    res["ctl_none"] = flat._ctl is None                       # the SAME verdict on both ranks (rank 0 did get a group)
    extra = nn.Linear(4, 1)
    x = torch.randn(3, 4)
    # degraded mode: the device-side exchange at the first step still refuses differing gradient sets on every rank
    (flat(x).sum() + (extra(x).sum() if rank == 1 else 0.0)).backward()
    params = flat.grad_params() + list(extra.parameters())
    try:
        flat.allreduce_grads(params)
        res["refused"] = False
    except RuntimeError as e:
        res["refused"] = "gradient sets differ" in str(e)
    for p in params:
        p.grad = None
    (flat(x).sum() + extra(x).sum()).backward()
    res["n_coll"] = flat.allreduce_grads(params)
    ddp.control_group = real
    # the cache holds the data-group OBJECT: after a teardown and a new job in this process the old entry is not handed out
    c0 = real()
    res["cached"] = real() is c0
    dist.barrier()
    dist.destroy_process_group()
    os.environ["MASTER_PORT"] = str(port + 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c1 = real()
    res["fresh_after_reinit"] = c1 is not c0 and c1 is dist.group.WORLD
    t = torch.ones(1)
    dist.all_reduce(t, group=c1)
    res["works"] = int(t.item()) == world
    torch.save(res, f"{out}.rank{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_control_group_verdict_is_collective_and_the_cache_survives_a_reinit(tmp_path):
    """ADVICE r5 (learning/ddp.py): creating the control group is a collective that may fail on ONE rank; that rank used to skip the
    per-step agreement while the others blocked in it.  Now the ranks agree on whether they all hold one (MIN over the data
    group); without it the gradient sets are still verified (device-side, first step / local change).  The cache is keyed on the
    data-group object, not on an id() that a later job's group can reuse."""
    out = str(tmp_path / "ctl")
    port = _free_port()
    mp.spawn(_ctl_fail_worker, args=(2, port, out), nprocs=2, join=True)
    for r in (torch.load(f"{out}.rank0"), torch.load(f"{out}.rank1")):
        assert r["ctl_none"] and r["refused"] and r["n_coll"] == 1
        assert r["cached"] and r["fresh_after_reinit"] and r["works"]
