"""Host-side logic that needs no GPU: the epoch learning-rate schedule against the reference's values, the projection
heads of CMO (all four kinds, reference state-dict keys), macro-F1, the bounded EMA-table cache."""
import argparse
import math

import os

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opt(**kw):
    base = dict(learning_rate=0.05, lr_decay_epochs=[30, 40, 60], lr_decay_rate=0.1, cosine=False, epochs=60)
    base.update(kw)
    return argparse.Namespace(**base)


def _ref_lr(epoch, opt):
    """reference helper/util.py:37-50 restated (values, not code)"""
    lr = opt.learning_rate
    if opt.cosine:
        eta_min = lr * opt.lr_decay_rate ** 3
        return eta_min + (lr - eta_min) * (1 + math.cos(math.pi * epoch / opt.epochs)) / 2
    steps = sum(epoch > e for e in opt.lr_decay_epochs)
    return lr * opt.lr_decay_rate ** steps if steps else lr


@pytest.mark.parametrize("cosine", [False, True])
def test_adjust_learning_rate_matches_reference_schedule(cosine):
    from moma_amd.helper.util import adjust_learning_rate
    opt = _opt(cosine=cosine)
    m = nn.Linear(2, 2)
    optim = torch.optim.SGD([{"params": [m.weight]}, {"params": [m.bias], "lr": 0.7}], lr=opt.learning_rate)
    for epoch in [1, 2, 15, 30, 31, 40, 41, 59, 60]:
        lr = adjust_learning_rate(epoch, opt, optim)
        assert lr == pytest.approx(_ref_lr(epoch, opt), rel=1e-12)
        assert all(g["lr"] == lr for g in optim.param_groups)          # EVERY group is rewritten each epoch
    if cosine:                                                          # known answers: start, middle, end of the cosine
        assert adjust_learning_rate(0, opt, optim) == pytest.approx(0.05)
        assert adjust_learning_rate(30, opt, optim) == pytest.approx((0.05 + 0.05e-3) / 2)
        assert adjust_learning_rate(60, opt, optim) == pytest.approx(0.05e-3)


@pytest.mark.parametrize("head,keys", [
    ("None", []),
    ("linear", ["1.weight", "1.bias"]),
    ("mlp", ["1.weight", "1.bias", "3.weight", "3.bias"]),
    ("mlp_byol", ["1.weight", "1.bias", "2.weight", "2.bias", "2.running_mean", "2.running_var", "2.num_batches_tracked",
                  "4.weight", "4.bias"]),
])
def test_cmo_heads_all_kinds(head, keys):
    """reference MoMA/criterion_moco_att.py:254-305: module indices (state-dict keys), output shape, unit L2 norm."""
    from moma_amd.MoMA.criterion_moco_att import CMO
    opt = argparse.Namespace(head=head, s_dim=48, t_dim=40, feat_dim=48 if head == "None" else 16, attn="self")
    cmo = CMO(opt)
    assert sorted(k for k in cmo.embed_s.state_dict()) == sorted(keys)
    x = torch.randn(6, 48, 1, 1)
    y = cmo.embed_s(x)
    assert y.shape == (6, opt.feat_dim)
    torch.testing.assert_close(y.norm(dim=1), torch.ones(6), rtol=1e-5, atol=1e-5)
    assert cmo.embed_t(torch.randn(6, 40, 1, 1)).shape == (6, 40 if head == "None" else 16)
    assert {n.split(".")[0] for n, _ in cmo.named_parameters() if n.startswith("atts")} == {"atts_q", "atts_k", "atts_queue"}


def test_macro_f1_known_answer():
    from moma_amd.helper.loops_moma import macro_f1
    cm = np.array([[5, 1, 0], [2, 3, 0], [0, 0, 0]])                    # class 2 never predicted right -> counts 0
    f0 = 2 * (5 / 7) * (5 / 6) / ((5 / 7) + (5 / 6))
    f1 = 2 * (3 / 4) * (3 / 5) / ((3 / 4) + (3 / 5))
    assert macro_f1(cm) == pytest.approx((f0 + f1) / 3)


def test_graphed_inference_contract_on_cpu_is_eager():
    """On CPU tensors (or with grad enabled) GraphedInference is an eager call under the SAME return contract as a replay:
    ([pooled feature], logits); full_feats=True hands out the whole feature list."""
    from moma_amd.helper.graphs import GraphedInference
    from moma_amd.backbones.resnet_cifar import resnet8
    m = resnet8(num_classes=10).eval()
    g = GraphedInference(m)
    x = torch.randn(2, 3, 32, 32)
    with torch.no_grad():
        feats, logits = g(x, is_feat=True)
        ref_feats, ref_logits = m(x, is_feat=True)
        all_feats, _ = g(x, is_feat=True, full_feats=True)
    assert len(feats) == 1 and torch.equal(feats[0], ref_feats[-1]) and torch.equal(logits, ref_logits)
    assert len(all_feats) == len(ref_feats)


def test_hip_env_switch_is_set_at_import_and_respects_the_caller():
    """moma_amd/hip_env.py: importing the package sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (before the process's first HIP call); a
    value the caller exported is left alone and configure() reports that the switch is not in place."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, moma_amd; from moma_amd import hip_env; print(os.environ[hip_env.SWITCH], hip_env.configure())"
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["0", "True"]
    env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "1"
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["1", "False"]


def test_loop_refuses_the_dual_queue_memories_up_front():
    """--mem MoCoST / MoCoSSTT: the reference's loop passes (q, k, all_k) to every memory (helper/loops_moma.py:331) and dies in its
    first step on these two (no k_t is ever computed).  Here the step object says so when it is built, before any compute."""
    import argparse
    import pytest
    import torch
    from moma_amd.helper.loops_moma import MomaStep

    class Dual:
        memory_s = None
    with pytest.raises(NotImplementedError, match="k_t"):
        MomaStep([None, None], [None, None, None], None, Dual(), None, argparse.Namespace(distill="moma"), torch.device("cpu"))
    MomaStep([None, None], [None, None, None], None, Dual(), None, argparse.Namespace(distill="kd"), torch.device("cpu"))   # (kd: no memory used)


def _launch_with_children(monkeypatch, n, child_code, argv=("--gpus", "2")):
    """Run bench.launch_ranks(n) with every rank replaced by `python -c child_code` (real processes, pipes and exit codes) and
    with every way of counting devices through HIP turned into an error."""
    import subprocess
    import sys
    import bench

    def refuse(*a, **k):
        raise AssertionError("the parent of the ranks asked the HIP runtime for devices")
    monkeypatch.setattr(torch._C, "_cuda_getDeviceCount", refuse)
    monkeypatch.setattr(torch.cuda, "device_count", refuse)
    monkeypatch.setattr(torch.cuda, "is_available", refuse)
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: n)
    monkeypatch.setattr(sys, "argv", ["bench.py", *argv])
    real, seen = subprocess.Popen, []

    _launch_with_children.procs = []

    def popen(cmd, **kw):
        seen.append((cmd, kw["env"]))
        _launch_with_children.procs.append(real([sys.executable, "-c", child_code], **kw))
        return _launch_with_children.procs[-1]
    monkeypatch.setattr(subprocess, "Popen", popen)
    rc = bench.launch_ranks(n)
    assert not torch.cuda.is_initialized()
    return rc, seen


def test_launch_ranks_parent_stays_off_the_gpu_and_starts_the_ranks_itself(monkeypatch, capfd):
    """bench.py --gpus N without a launcher (ADVICE r4 / VERDICT r4 item 5): the parent counts devices from sysfs only, starts the
    N ranks directly (no torch.distributed.run, whose launcher opens the device) with the rendezvous variables torch.distributed's
    env:// init reads, relays rank 0's one JSON line, and returns 0."""
    import json
    import os
    code = ("import os, json; r = os.environ['RANK']; print('progress from rank', r); "
            "print(json.dumps({k: os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))")
    rc, seen = _launch_with_children(monkeypatch, 3, code, argv=("--gpus", "3", "--steps", "2"))
    out, err = capfd.readouterr()
    assert rc == 0 and len(seen) == 3
    for r, (cmd, env) in enumerate(seen):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "3", "--steps", "2"] and "torch.distributed.run" not in cmd
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "3", "127.0.0.1")
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        assert "MOMA_BENCH_SELF_LAUNCH" not in env
    assert len({env["MASTER_PORT"] for _, env in seen}) == 1
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["RANK"] == "0"          # stdout = rank 0's JSON line, nothing else
    assert "progress from rank 0" in err and "progress from rank 2" in err


def test_launch_ranks_a_failed_rank_ends_the_job(monkeypatch, capfd):
    """One rank dies: the others (which would wait in a collective forever) are terminated and the parent reports the failure."""
    import time
    code = "import os, sys, time; sys.exit(5) if os.environ['RANK'] == '1' else time.sleep(600)"
    t0 = time.time()
    rc, _ = _launch_with_children(monkeypatch, 2, code)
    assert rc == 5 and time.time() - t0 < 60
    assert "rank 1 exited with code 5" in capfd.readouterr().err


def test_launch_ranks_deadline_stops_stuck_ranks_and_says_how_far_they_got(monkeypatch, capfd, tmp_path):
    """eight ranks parked in init_process_group (or a collective) must not sit there until the
    driver's own timeout and leave nothing.  Past --launch_timeout the parent stops every rank -- SIGTERM, then SIGKILL for one
    that ignores it --, exits 124 and prints, per rank, the phases it reached and the tail of its stderr (also kept as
    bench_rank<r>.err)."""
    import time
    code = ("import os, sys, time, signal\n"
            "r = os.environ['RANK']\n"
            "open(os.environ['MOMA_BENCH_PHASE_FILE'], 'a').write('started %f\\n' % time.time())\n"
            "if r == '1':\n"
            "    signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"            # a rank stuck where SIGTERM does not reach
            "    open(os.environ['MOMA_BENCH_PHASE_FILE'], 'a').write('rccl_init %f\\n' % time.time())\n"
            "print('rank', r, 'waiting in a collective', file=sys.stderr, flush=True)\n"
            "time.sleep(600)\n")
    monkeypatch.setenv("MOMA_BENCH_LOG_DIR", str(tmp_path))
    monkeypatch.setenv("MOMA_BENCH_LAUNCH_TIMEOUT", "3")
    t0 = time.time()
    rc, seen = _launch_with_children(monkeypatch, 2, code)
    took = time.time() - t0
    err = capfd.readouterr().err
    assert rc == 124 and took < 40, (rc, took)
    assert "deadline of 3 s reached" in err and "rank 0: started" in err and "rank 1: started" in err and "-> rccl_init" in err
    assert "[rank 1 stderr] rank 1 waiting in a collective" in err
    assert all(env["NCCL_DEBUG"] == "WARN" and env["MOMA_BENCH_PHASE_FILE"].endswith(f"bench_rank{r}.phase") for r, (_, env) in enumerate(seen))
    assert "waiting in a collective" in (tmp_path / "bench_rank0.err").read_text()
    # nothing left behind: both children are gone (the SIGTERM-deaf one was killed)
    assert all(p.poll() is not None for p in _launch_with_children.procs)
    assert _launch_with_children.procs[1].returncode == -9


def test_launch_ranks_reports_the_phases_when_a_rank_fails(monkeypatch, capfd, tmp_path):
    code = ("import os, sys, time\n"
            "f = os.environ['MOMA_BENCH_PHASE_FILE']\n"
            "open(f, 'a').write('started %f\\nrccl_init %f\\n' % (time.time(), time.time()))\n"
            "if os.environ['RANK'] == '0':\n"
            "    open(f, 'a').write('wrap_self_test %f\\n' % time.time()); print('RuntimeError: self-test', file=sys.stderr); sys.exit(7)\n"
            "time.sleep(600)\n")
    monkeypatch.setenv("MOMA_BENCH_LOG_DIR", str(tmp_path))
    rc, _ = _launch_with_children(monkeypatch, 2, code)
    err = capfd.readouterr().err
    assert rc == 7 and "rank 0 exited with code 7" in err
    assert "rank 0: started" in err and "-> wrap_self_test" in err and "rank 1: started" in err and "[rank 0 stderr] RuntimeError: self-test" in err


def test_rank_watchdog_reports_phase_and_stacks_then_exits_124():
    """Inside a rank (so also under torch.distributed.run, how the driver starts its N > 1 runs): past its deadline a stuck rank
    says which phase it is in, dumps the stacks of its threads and exits 124."""
    import subprocess
    import sys
    code = ("import sys, time; sys.argv = ['bench.py']; sys.path.insert(0, %r); import bench\n"
            "bench.rank_watchdog(1.0); bench.phase('started'); bench.phase('rccl_init'); time.sleep(60)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, RANK="3", WORLD_SIZE="8"))
    assert r.returncode == 124, r.stderr[-2000:]
    assert "rank 3/8 phase rccl_init" in r.stderr and "rank 3: deadline of 1 s reached in phase 'rccl_init'" in r.stderr
    assert "started +0s -> rccl_init +0s" in r.stderr and "most recent call first" in r.stderr


def test_cli_parent_of_the_ranks_makes_no_hip_call(monkeypatch):
    """`train_student_moma.py --multiprocessing-distributed` (the reference's launch mode, train_student_moma.py:207-224): the
    process in front of mp.spawn counts devices from sysfs and never asks the HIP runtime (VERDICT r5 missing #2: it called
    torch.cuda.device_count(), which ends in hipGetDeviceCount on this platform)."""
    import torch.multiprocessing as mp
    import moma_amd.devices as devices
    import moma_amd.train_student_moma as tsm

    def refuse(*a, **k):
        raise AssertionError("the parent of the ranks asked the HIP runtime for devices")
    monkeypatch.setattr(torch._C, "_cuda_getDeviceCount", refuse)
    monkeypatch.setattr(torch.cuda, "device_count", refuse)
    monkeypatch.setattr(torch.cuda, "is_available", refuse)
    monkeypatch.setattr(devices, "visible_gpu_count", lambda: 3)
    spawned = []
    monkeypatch.setattr(mp, "spawn", lambda fn, nprocs, args: spawned.append((fn, nprocs, args)))
    tsm.main(["--distill", "moma", "--multiprocessing-distributed", "--gpu_id", "0,1,2", "--model_s", "resnet8x4", "--model_t", "resnet8x4"])
    (fn, nprocs, (ngpus, opt)), = spawned
    assert fn is tsm.main_worker and nprocs == ngpus == opt.world_size == 3
    assert os.environ["CUDA_VISIBLE_DEVICES"] == "0,1,2" and not torch.cuda.is_initialized()
    # no KFD sysfs to read: the length of the user's --gpu_id list
    monkeypatch.setattr(devices, "visible_gpu_count", lambda: None)
    assert tsm.parent_gpu_count("0,1") == 2
    monkeypatch.setattr(devices, "visible_gpu_count", lambda: 0)
    with pytest.raises(SystemExit):
        tsm.parent_gpu_count("0")


def test_launch_ranks_refuses_more_ranks_than_gpus(monkeypatch):
    import bench
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 1)
    monkeypatch.delenv("MOMA_BENCH_SAME_DEVICE", raising=False)
    assert bench.launch_ranks(2) == 2


def test_visible_gpu_count_reads_sysfs_and_the_visibility_masks(monkeypatch, tmp_path):
    """KFD topology: nodes with simd_count > 0 are GPUs (CPU nodes have 0); *_VISIBLE_DEVICES narrow the count; no sysfs = None."""
    import glob
    import bench
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    real = glob.glob
    monkeypatch.setattr(glob, "glob", lambda pat: real(str(tmp_path / "*" / "properties")) if "kfd" in pat else real(pat))
    # (no drm_render_minor line: counted; with one, the render node must exist and be accessible)
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.visible_gpu_count() == 3
    with open(tmp_path / "4" / "properties", "a") as f:
        f.write("drm_render_minor 99999\n")                     # a GPU of the host that this container was not given
    assert bench.visible_gpu_count() == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpu_count() == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2,3")
    assert bench.visible_gpu_count() == 2
    monkeypatch.setattr(glob, "glob", lambda pat: [])
    assert bench.visible_gpu_count() is None


def test_fused_sgd_keeps_weight_versions_moving_and_starts_from_zero_momentum():
    """train_student_moma.make_optimizer, --amp fp16 path (torch's fused SGD; forced onto the CPU here): (1) a step advances the
    parameters' version counters -- torch._fused_sgd_ itself does not, and the attention weight packs are keyed on them; (2) the
    momentum buffers exist before the first step, as zeros, so a first step that the GradScaler skips (found_inf) cannot leave
    them uninitialised; with dampening 0 the first real step equals the plain optimizer's."""
    import moma_amd.train_student_moma as tsm
    torch.manual_seed(0)
    w0 = torch.randn(5, 3)
    g = [torch.randn(5, 3), torch.randn(5, 3)]
    opt = argparse.Namespace(amp="fp16", learning_rate=0.1, momentum=0.9, weight_decay=1e-2)
    a = nn.Parameter(w0.clone())
    fused = tsm.make_optimizer([a], opt, "cpu", fused=True)
    assert fused.defaults.get("fused") and torch.equal(fused.state[a]["momentum_buffer"], torch.zeros(5, 3))
    b = nn.Parameter(w0.clone())
    plain = torch.optim.SGD([b], lr=0.1, momentum=0.9, weight_decay=1e-2)
    # a skipped first step (found_inf = 1): nothing moves, the buffer stays zero -- not garbage
    a.grad = g[0].clone()
    fused.grad_scale, fused.found_inf = torch.tensor(1.0), torch.tensor(1.0)
    fused.step()
    del fused.grad_scale, fused.found_inf
    assert torch.equal(a.detach(), w0) and torch.equal(fused.state[a]["momentum_buffer"], torch.zeros(5, 3))
    v0 = a._version
    for gi in g:
        a.grad, b.grad = gi.clone(), gi.clone()
        fused.step(); plain.step()
        assert torch.allclose(a.detach(), b.detach(), rtol=0, atol=1e-6)
    assert a._version >= v0 + 2
