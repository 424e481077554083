"""End-to-end CLI on the GPU: `train_student_moma.py --distill moma` with the reference's flags on synthetic data
(1 process, 1 GPU), fused and reference-sequence KD paths; checks it trains, checkpoints and logs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--no_fused", "--moma_prec", "fp32", "--queue_dtype", "bf16"]])
def test_cli_trains_on_synthetic(tmp_path, extra):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "train_student_moma.py"), "--distill", "moma", "--model_s", "resnet8x4",
           "--model_t", "resnet8x4", "--dataset", "cifar100", "--n_cls", "2", "--batch_size", "32", "--epochs", "2",
           "--steps_per_epoch", "6", "--nce_k", "1024", "--head", "mlp", "--feat_dim", "128", "-c", "1", "-d", "1", "-b", "1",
           "--print_freq", "3", "--miopen_find", "off", "--save_root", str(tmp_path)] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "images/sec" in r.stdout and "best accuracy" in r.stdout
    saved = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs]
    assert any(f.endswith("net_best_acc.pth") for f in saved) and any(f.endswith("net_best_f1.pth") for f in saved)
    stat = [f for f in saved if f.endswith("stat.json")]
    st = json.load(open(stat[0]))                                       # per-epoch confusion matrices (reference :573-591)
    assert sorted(st) == ["1", "2"] and np.array(st["2"]["val_cf"]).shape == (2, 2)
    assert abs(np.trace(np.array(st["2"]["val_cf"])) / np.sum(st["2"]["val_cf"]) * 100 - st["2"]["val_acc"]) < 1e-6
    params = [f for f in saved if f.endswith("parameters.json")]
    assert params and json.load(open(params[0]))["nce_t"] == 0.15       # forced for --distill moma (reference :135)


@pytest.mark.parametrize("amp", [None, "fp16"])
def test_cli_resume_continues_queue_pointer(tmp_path, amp):
    """the trainer's own checkpoint carries queue + pointer + CMO + EMA teacher; --resume continues.
    amp = fp16: the fused SGD's state (momentum buffers created up front) and the GradScaler's scale travel too."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    base = [sys.executable, os.path.join(ROOT, "train_student_moma.py"), "--distill", "moma", "--model_s", "resnet8x4",
            "--model_t", "resnet8x4", "--dataset", "cifar100", "--n_cls", "2", "--batch_size", "32", "--steps_per_epoch", "5",
            "--nce_k", "1024", "--head", "mlp", "--feat_dim", "128", "-c", "1", "-d", "1", "-b", "1", "--miopen_find", "off", "--save_root", str(tmp_path)]
    if amp:
        base += ["--amp", amp]
    r = subprocess.run(base + ["--epochs", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ck = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f == "ckpt_last.pth"]
    assert ck
    state = torch.load(ck[0], map_location="cpu")
    assert state["contrast"]["_extra_state"]["index"] == (5 * 32) % 1024 and state["epoch"] == 1
    r = subprocess.run(base + ["--epochs", "2", "--resume", ck[0]], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "resumed from" in r.stdout and "queue pointer 160" in r.stdout
    state2 = torch.load(ck[0], map_location="cpu")
    assert state2["epoch"] == 2 and state2["contrast"]["_extra_state"]["index"] == (10 * 32) % 1024
    # per-rank state (this rank's queue + pointer, RNG streams) next to the shared file
    mine = os.path.join(os.path.dirname(ck[0]), "ckpt_last_rank0.pth")
    rs = torch.load(mine, map_location="cpu", weights_only=False)
    assert "per-rank state found" in r.stdout
    assert rs["contrast"]["_extra_state"]["index"] == (10 * 32) % 1024 and set(rs["rng"]) == {"python", "numpy", "torch", "cuda"}
    if amp == "fp16":
        # the scale after epoch 1 is what epoch 2 starts from (a fresh scaler would start at 2^16 again and could only be <= it)
        assert state["grad_scaler"]["scale"] <= 65536.0 and state2["grad_scaler"]["scale"] <= state["grad_scaler"]["scale"]
        assert state2["grad_scaler"]["_growth_tracker"] >= state["grad_scaler"]["_growth_tracker"] or \
            state2["grad_scaler"]["scale"] < state["grad_scaler"]["scale"]
        assert all(v["momentum_buffer"] is not None for v in state2["optimizer"]["state"].values())
    else:
        assert state.get("grad_scaler") is None


def test_cli_reproducible_mode_trains(tmp_path):
    """`--reproducible` = the reference's default semantics (a seed + cudnn.deterministic, train_student_moma.py:241-246): MIOpen's
    deterministic algorithms and no find mode.  What it buys is measured elsewhere (tests/test_gpu_step_graph.py: inside one process
    the loop repeats bit for bit -- nothing in this library adds in an unordered way; ACROSS processes MIOpen's own solver choice
    can still differ: scripts/diag_cli_trace.py saw 2-3 distinct traces over 5 processes, parting at the second step, graphs on or
    off); here: the mode is accepted, trains with the step served from graphs, and says what it is."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "train_student_moma.py"), "--distill", "moma", "--model_s", "resnet8x4",
           "--model_t", "resnet8x4", "--dataset", "cifar100", "--n_cls", "2", "--batch_size", "32", "--epochs", "2",
           "--steps_per_epoch", "7", "--nce_k", "1024", "--head", "mlp", "--feat_dim", "128", "-c", "1", "-d", "1", "-b", "1",
           "--reproducible", "--save_root", str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "reproducible: MIOpen deterministic algorithms" in r.stdout and "best accuracy" in r.stdout
    ck = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f == "ckpt_last.pth"]
    state = torch.load(ck[0], map_location="cpu", weights_only=False)
    assert state["epoch"] == 2 and state["contrast"]["_extra_state"]["index"] == (14 * 32) % 1024
    assert all(torch.isfinite(v).all() for v in state["model"].values() if v.is_floating_point())


def test_cli_reference_launch_mode_spawns_one_rccl_rank_and_resumes(tmp_path):
    """The reference's OWN launch mode for `--distill moma` (train_student_moma.py:207-224: --multiprocessing-distributed, mp.spawn,
    one process per GPU, backend 'nccl' = RCCL): on the one GPU of the box that is ONE spawned rank on a real RCCL communicator
    -- init_ddp_environment, the student wrap, broadcast_memory, the gradient all-reduce and the epoch-end metric all-reduce all
    run.  Two epochs, then a resumed third: the queue pointer continues, the per-rank state file is found."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    base = [sys.executable, os.path.join(ROOT, "train_student_moma.py"), "--distill", "moma", "--multiprocessing-distributed",
            "--dist-url", f"tcp://127.0.0.1:{_free_port()}", "--gpu_id", "0", "--model_s", "resnet8x4", "--model_t", "resnet8x4",
            "--dataset", "cifar100", "--n_cls", "2", "--batch_size", "32", "--steps_per_epoch", "5", "--nce_k", "1024", "--head", "mlp",
            "--feat_dim", "128", "-c", "1", "-d", "1", "-b", "1", "--print_freq", "2", "--miopen_find", "off", "--save_root", str(tmp_path)]
    r = subprocess.run(base + ["--epochs", "2"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Use GPU: 0 for training" in r.stdout and "best accuracy" in r.stdout and "images/sec" in r.stdout
    ck = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f == "ckpt_last.pth"]
    assert ck
    state = torch.load(ck[0], map_location="cpu")
    assert state["epoch"] == 2 and state["contrast"]["_extra_state"]["index"] == (10 * 32) % 1024
    r = subprocess.run(base + ["--epochs", "3", "--resume", ck[0]], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "resumed from" in r.stdout and "queue pointer 320" in r.stdout and "per-rank state found" in r.stdout
    state = torch.load(ck[0], map_location="cpu")
    assert state["epoch"] == 3 and state["contrast"]["_extra_state"]["index"] == (15 * 32) % 1024


def test_cli_distill_kd_default_path(tmp_path):
    """`--distill kd` is the CLI default (reference train_student_moma.py:91): CE + KL only, no ContrastTrainer; the
    graphed teacher must work without one (round-1 ADVICE: AttributeError on NoneType)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "train_student_moma.py"), "--distill", "kd", "--model_s", "resnet8x4",
           "--model_t", "resnet8x4", "--dataset", "cifar100", "--n_cls", "2", "--batch_size", "32", "--epochs", "1",
           "--steps_per_epoch", "6", "--print_freq", "3", "--miopen_find", "off", "--cosine", "--save_root", str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "images/sec" in r.stdout and "best accuracy" in r.stdout


@pytest.mark.parametrize("head", ["linear", "mlp"])
def test_cli_config1_resnet8x4_student_resnet32x4_teacher(tmp_path, head):
    """BASELINE configs[0]'s model pair: resnet8x4 student / resnet32x4 teacher, CIFAR-shaped inputs, B = 8.  Different depths ->
    the reference's zip-EMA raises half-way; defined behaviour here: the teacher stays frozen.  Also covers
    the `linear` head (reference MoMA/criterion_moco_att.py:254-305)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, os.path.join(ROOT, "train_student_moma.py"), "--distill", "moma", "--model_s", "resnet8x4",
           "--model_t", "resnet32x4", "--dataset", "cifar100", "--n_cls", "100", "--batch_size", "8", "--epochs", "1",
           "--steps_per_epoch", "6", "--nce_k", "1024", "--head", head, "--feat_dim", "128", "-c", "1", "-d", "1", "-b", "1",
           "--print_freq", "3", "--miopen_find", "off", "--save_root", str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "teacher stays frozen" in r.stdout and "images/sec" in r.stdout
    params = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f.endswith("parameters.json")]
    p = json.load(open(params[0]))
    assert p["s_dim"] == 256 and p["t_dim"] == 256 and p["feat_dim"] == 128


def _run_cli(tmp_path, args, timeout=900):
    cmd = [sys.executable, os.path.join(ROOT, "train_student_moma.py"), "--distill", "moma", "--dataset", "synthetic",
           "-c", "1", "-d", "1", "-b", "1", "--print_freq", "2", "--miopen_find", "off", "--save_root", str(tmp_path)] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_cli_config3_vit_small_pair_8_heads(tmp_path):
    """BASELINE configs[2]: ViT-S student + teacher, 8-head attention-KD path, d = 384 (--head None -> hd = 48), bf16.
    (The reference has no runnable ViT backbone, SURVEY Q13: the backbone is the build's own; the KD term is the
    same kernels -- fused K1 core with hd = 48 and the one-pass K2 with d = 384.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = _run_cli(tmp_path, ["--model_s", "vit_small_patch16_224", "--model_t", "vit_small_patch16_224", "--n_cls", "2",
                            "--image_size", "64", "--batch_size", "32", "--epochs", "1", "--steps_per_epoch", "6",
                            "--nce_k", "4096", "--head", "None", "--num_heads", "8", "--amp", "bf16",
                            "--queue_dtype", "bf16"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "images/sec" in r.stdout
    params = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f.endswith("parameters.json")]
    p = json.load(open(params[0]))
    assert p["s_dim"] == 384 and p["t_dim"] == 384 and p["feat_dim"] == 384 and p["num_heads"] == 8


def test_cli_config5_cross_arch_vit_base_to_resnet50_fp16(tmp_path):
    """BASELINE configs[4]: ViT-B teacher -> ResNet-50 student, --head mlp (s_dim 2048, t_dim 768 -> d = 512), fp16
    autocast, teacher checkpoint loaded non-strictly (--tec_strict).  Cross-architecture: the reference's zip-EMA
    raises half-way; defined behaviour here = teacher (and its head) stay frozen, training proceeds."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from moma_amd.backbones import model_dict
    torch.manual_seed(0)
    sd = model_dict["vit_base_patch16_224"](num_classes=7).state_dict()      # a locally saved teacher with another head
    ck = os.path.join(tmp_path, "teacher.pth")
    torch.save({"model": sd}, ck)
    r = _run_cli(tmp_path, ["--model_s", "ResNet50", "--model_t", "vit_base_patch16_224", "--path_t", ck, "--tec_strict",
                            "--n_cls", "2", "--image_size", "64", "--batch_size", "16", "--epochs", "1",
                            "--steps_per_epoch", "4", "--nce_k", "2048", "--head", "mlp", "--feat_dim", "512",
                            "--amp", "fp16", "--queue_dtype", "bf16"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "teacher stays frozen" in r.stdout and "images/sec" in r.stdout
    params = [os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f.endswith("parameters.json")]
    p = json.load(open(params[0]))
    assert p["s_dim"] == 2048 and p["t_dim"] == 768 and p["feat_dim"] == 512


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


# (the bench's own learning rate, 0.05.  Round 4 ran these at 0.01 after one graph-served rehearsal ended in NaN; the cause found an
#  hour later was the runtime's graph packet capture dropping captured memset nodes -- garbage bias gradients of nn.Linear layers,
#  moma_amd/hip_env.py + helper/graphs.py:replay_is_safe() -- not the miniature's learning rate, so the guard is back on the
#  configuration that failed.)
_SMALL = ["--steps", "6", "--warmup", "5", "--batch_size", "32", "--image_size", "64", "--nce_k", "4096", "--no_cpu_baseline"]


def _check_two_rank_line(out, dp, steps=6, warmup=5):
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["scaling"] == "weak"
    d = out["dist"]
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["overlap_teacher"] and d["graph_teacher"]
    assert d["dp_wrap"] == {"flat": "FlatDataParallel", "ddp": "DistributedDataParallel"}[dp]
    assert d["criterion_allreduce_launches"] == warmup + steps
    assert d["replica_checksum_spread"] == {"student": 0.0, "criterion": 0.0, "ema_teacher": 0.0}, d
    # per-rank view of the timed region (what an N = 8 line that misses its efficiency target has to explain itself with)
    for key in ("per_rank_ms_per_step_median", "per_rank_host_issue_ms_median", "per_rank_allreduce_grads_ms",
                "per_rank_buffer_broadcast_ms", "per_rank_loss"):
        assert len(d[key]) == 2, key
    assert all(v > 0 for v in d["per_rank_ms_per_step_median"] + d["per_rank_host_issue_ms_median"] + d["per_rank_buffer_broadcast_ms"])
    assert all(np.isfinite(v) for v in d["per_rank_loss"]) and np.isfinite(out["loss_mean_timed_steps"])
    if dp == "flat":
        assert all(v > 0 for v in d["per_rank_allreduce_grads_ms"])
    else:
        assert d["per_rank_allreduce_grads_ms"] == [0.0, 0.0]          # the stock reducer's buckets: not under these events


@pytest.mark.parametrize("dp", ["flat", "ddp"])
def test_bench_two_ranks_on_one_gpu_replicas_stay_in_sync(tmp_path, dp):
    """The combination the driver's multi-GPU run hits first, rehearsed on ONE GPU: `python -m torch.distributed.run` with two
    ranks -> bench.py --gpus 2 (gloo carries the collectives: RCCL refuses two ranks on one device) with the HIP-graphed teacher
    and the teacher side on a second stream ON, under both student wraps: dp = flat -- learning/ddp.py:FlatDataParallel, ONE flat
    gradient all-reduce per step behind the backward + one flat buffer broadcast per forward; dp = ddp -- the stock reducer with
    the criterion gradients' flat all-reduce launched from autograd hooks.  After warm-up and timed steps the replicas must be
    BIT-identical (student, criterion modules, EMA teacher) and the JSON line must carry the N > 1 fields."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, MOMA_BENCH_SAME_DEVICE="1", MOMA_BENCH_BACKEND="gloo", MOMA_BENCH_FORCE_OVERLAP="1", MOMA_DP=dp)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + _SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    _check_two_rank_line(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), dp)


def test_bench_gpus2_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment starts its own two ranks (before any GPU call in the parent,
    which only relays rank 0's JSON line and the exit code) -- as the reference's entry point spawns its ranks itself
    (train_student_moma.py:215-224).  Default wrap (MOMA_DP unset = auto: the collectives' self-test picks the flat wrap)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MOMA_DP")}
    env.update(MOMA_BENCH_SAME_DEVICE="1", MOMA_BENCH_BACKEND="gloo", MOMA_BENCH_FORCE_OVERLAP="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + _SMALL, capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines          # stdout carries exactly the one JSON line
    _check_two_rank_line(json.loads(lines[0]), "flat")
    assert "starting 2 ranks" in r.stderr


def test_bench_self_launch_parent_path_on_one_rccl_rank(tmp_path):
    """The code `python bench.py --gpus N` runs on the driver's 8-GPU node, with the one rank a one-GPU box allows and NO
    MOMA_BENCH_SAME_DEVICE: the parent counts devices from sysfs (never through HIP), starts the rank itself (no
    torch.distributed.run in between) and relays its JSON line; the rank initialises RCCL ("nccl") from the env:// variables the
    parent set and runs the N > 1 step (flat wrap, step graphs, side stream).  MOMA_BENCH_SELF_LAUNCH=1 sends --gpus 1 down
    that path; MOMA_BENCH_FORCE_DIST=1 makes the one rank build its process group."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MOMA_DP", "MOMA_BENCH_SAME_DEVICE",
                                                            "MOMA_BENCH_BACKEND")}
    env.update(MOMA_BENCH_SELF_LAUNCH="1", MOMA_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + _SMALL, capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["dist"]["backend"] == "nccl" and out["dist"]["world_size"] == 1
    assert out["dist"]["dp_wrap"] == "FlatDataParallel" and np.isfinite(out["loss_mean_timed_steps"])
    assert out["config"]["step_graphs"]["timed_steps_replayed"] >= 4, out["config"]["step_graphs"]
    assert "starting 1 ranks" in r.stderr and "torch.distributed.run" not in r.stderr.split("starting 1 ranks")[1].splitlines()[0]
    # and more ranks than the box has GPUs is refused by the parent, from sysfs, before anything starts
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + _SMALL, capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env={k: v for k, v in env.items() if not k.startswith("MOMA_BENCH")})
    assert r.returncode == 2 and "only 1 GPU(s) visible" in r.stderr, (r.returncode, r.stderr[-2000:])


def test_bench_refuses_a_non_finite_loss(tmp_path):
    """A timed region whose loss is not finite measured a broken step: bench.py prints no result line and exits non-zero (round 3:
    a development run with `Loss nan` produced a normal-looking JSON line).  Forced here by a learning rate of 1e30."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "3", "--batch_size", "32",
                        "--image_size", "64", "--nce_k", "4096", "--no_cpu_baseline", "--learning_rate", "1e30"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 3, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")] and "non-finite loss" in r.stderr


@pytest.mark.parametrize("dp", ["flat", "ddp"])
def test_bench_one_rank_on_rccl(tmp_path, dp):
    """The N > 1 code path on RCCL ITSELF with the one rank a one-GPU box allows: torchrun -> bench.py with
    MOMA_BENCH_FORCE_DIST=1 -> init_process_group("nccl") (communicator + watchdog thread live); dp = flat: the default wrap (one
    flat buffer broadcast per forward, ONE flat gradient all-reduce per step); dp = ddp: the stock reducer on the student with the
    hook-launched flat all-reduce of the criterion gradients issued on the same communicator; HIP-graph capture of the teacher
    and the side stream ON in both.  Must finish, report the dist fields with backend nccl, and time like the plain run does
    (no step an order of magnitude over the median: a capture that collides with the watchdog or a blocked collective shows up there)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    port = _free_port()
    env = dict(os.environ, MOMA_BENCH_FORCE_DIST="1", MOMA_DP=dp)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "5",
           "--batch_size", "64", "--image_size", "96", "--nce_k", "8192", "--no_cpu_baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    d = out["dist"]
    assert d["world_size"] == 1 and d["backend"] == "nccl" and d["overlap_teacher"] and d["graph_teacher"]
    assert d["dp_wrap"] == {"flat": "FlatDataParallel", "ddp": "DistributedDataParallel"}[dp]
    assert d["replica_checksum_spread"] == {"student": 0.0, "criterion": 0.0, "ema_teacher": 0.0}
    assert d["criterion_allreduce_launches"] == 5 + 8           # one flat all-reduce per warm-up and timed step
    # (a capture inside the timed region or a collective that blocks costs hundreds of milliseconds against a median of ~11;
    #  the bound was 3x until one step of one run took 3.02x on a busy box)
    assert out["ms_per_step_max"] < 10.0 * out["ms_per_step_median"], out
    assert "Loss nan" not in r.stderr
