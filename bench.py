#!/usr/bin/env python3
"""Headline benchmark: images/sec of the MoMA train step (EffNet-B0 pair, 224 px, K=65536) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full pass of helper/loops_moma.py's moma branch over one synthetic batch: student fwd,
teacher fwd x2 (Q8), CE + KL, multi-tensor EMA (K4), Shuffle-BN key encoding, heads, batch-token attention
x3 (K1), one-pass InfoNCE over the K x d queue (K2), ring enqueue (K3), backward, SGD step (and, for N>1,
the RCCL gradient all-reduce).  Inputs are resident in HBM before the timed region.  Weak scaling: per-rank
batch fixed, per-rank queue, no data-path collective besides the gradient all-reduce.

Rank 0 prints ONE JSON line with the driver's contract plus
  "roofline"     : the dominant hand-written kernel (K2, one-pass InfoNCE) -- algorithmic bytes/flops per
                   launch over its HIP-event-measured mean duration inside the timed region;
  "cpu_baseline" : the CPU restatement of the same step (oracle/step_oracle.py, pinned to the reference's
                   trace) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from moma_amd.miopen_env import use_shipped_db  # noqa: E402  (before torch: selects MIOpen's tuned solvers)
from moma_amd.devices import visible_gpu_count  # noqa: E402,F401  (sysfs only: the parent of the ranks never opens the device)

use_shipped_db(tag=os.environ.get("LOCAL_RANK", "0"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3    # f32-input MFMA (= the fp32 vector rate; MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=30)
    p.add_argument("--warmup", type=int, default=8)
    p.add_argument("--batch_size", type=int, default=256, help="per-rank batch")
    p.add_argument("--image_size", type=int, default=224)
    p.add_argument("--nce_k", type=int, default=65536)
    p.add_argument("--head", default="mlp", choices=["None", "linear", "mlp"])
    p.add_argument("--feat_dim", type=int, default=512)
    p.add_argument("--model", default="effiB0")
    p.add_argument("--model_t", default=None, help="teacher architecture (default: the student's; a different one keeps the "
                                                   "teacher frozen -- BASELINE configs[4]: --model ResNet50 --model_t vit_base_patch16_224)")
    p.add_argument("--n_cls", type=int, default=4)
    p.add_argument("--num_heads", type=int, default=4, help="heads of the batch-token attention modules (reference: 4; BASELINE configs[2]: 8)")
    p.add_argument("--moma_prec", default="bf16", choices=["fp32", "bf16"])
    p.add_argument("--queue_dtype", default="bf16", choices=["fp32", "bf16"])
    p.add_argument("--amp", default="bf16", choices=["none", "bf16", "fp16"])
    p.add_argument("--channels_last", action="store_true",
                   help="NHWC backbones (MIOpen's depthwise backward is ~7x slower in NHWC on gfx950: off by default)")
    p.add_argument("--learning_rate", type=float, default=0.05)
    p.add_argument("--print_freq", type=int, default=10 ** 9, help="loop print (device arithmetic + read-back) every N steps; default: only at step 0")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--cpu_batch", type=int, default=16)
    p.add_argument("--cpu_steps", type=int, default=15)    # ~10 s of host work at B=16
    p.add_argument("--miopen_find", action="store_true", help="cudnn.benchmark=True (MIOpen find mode)")
    p.add_argument("--no_graph_student", dest="graph_student", action="store_false",
                   help="issue the step launch by launch instead of replaying it from HIP graphs (helper/step_graph.py)")
    p.add_argument("--prefetch_queue", action="store_true",
                   help="experiment switch, OFF by default since round 5: a side-stream sweep that warms the Infinity Cache with the "
                        "queue ahead of K2 -- a second read of the 67 MB queue per step (15 us beside the attention launches) that "
                        "bought 0.6 us of the one-pass kernel (profiles/r04_step_kernel_stats.csv)")
    p.add_argument("--launch_timeout", type=float, default=None,
                   help="--gpus N without a launcher: seconds after which the parent stops every rank (SIGTERM, then SIGKILL), prints "
                        "how far each one got and exits 124 (default: MOMA_BENCH_LAUNCH_TIMEOUT or 420)")
    p.add_argument("--no_overlap_teacher", dest="overlap_teacher", action="store_false",
                   help="queue the teacher / key side of the step on the main stream instead of a second HIP stream")
    return p.parse_args()


class EventRecorder:
    """HIP events on the current stream around named C-ABI calls (moma_amd.ops.set_event_recorder)."""

    def __init__(self):
        self.enabled = False
        self.events = {}

    def __call__(self, name):
        rec = self

        class Ctx:
            def __enter__(self_inner):
                if rec.enabled:
                    self_inner.e0 = torch.cuda.Event(enable_timing=True)
                    self_inner.e1 = torch.cuda.Event(enable_timing=True)
                    self_inner.e0.record()
                return self_inner

            def __exit__(self_inner, *a):
                if rec.enabled:
                    self_inner.e1.record()
                    rec.events.setdefault(name, []).append((self_inner.e0, self_inner.e1))
                return False
        return Ctx()

    def mean_ms(self, name):
        ev = self.events.get(name, [])
        if not ev:
            return None
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)


class KernelEvents:
    """Raw hipEvent_t pairs handed to moma_infonce_fused_ex, which records them on the launch stream right
    before / after the one-pass kernel (the kernel the roofline is quoted for)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")          # torch's HIP runtime, already loaded in this process
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.pairs = []
        self.enabled = False

    def __call__(self):
        if not self.enabled:
            return (None, None, None)
        a, b, c = self.C.c_void_p(), self.C.c_void_p(), self.C.c_void_p()
        for e in (a, b, c):
            assert self.hip.hipEventCreate(self.C.byref(e)) == 0
        self.pairs.append((a, b, c))
        return (a.value, b.value, c.value)

    def mean_ms(self, last=1):
        """last = 1: begin .. end of the dominant kernel; last = 2: begin of the dominant kernel .. end of the call's last kernel."""
        if not self.pairs:
            return None
        tot = 0.0
        for t in self.pairs:
            ms = self.C.c_float()
            assert self.hip.hipEventElapsedTime(self.C.byref(ms), t[0], t[last]) == 0
            tot += ms.value
        return tot / len(self.pairs)


def _latest_profile(pattern):
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return fs[-1] if fs else None


def k2_pmc_traffic():
    """HBM bytes per launch of the one-pass kernel, and of the whole K2 call (one-pass kernel + combine; the queue pre-fetch sweep
    of rounds 1-4 is off), from the newest committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE runs of
    scripts/bench_k2.py on the bench shape, summarised in profiles/rNN_k2_pmc.csv): FETCH_SIZE is doubled per the guide's gfx950
    correction.  -> (kernel bytes, source, whole-call bytes or None); None when no summary is present.  The whole-call figure
    needs a summary collected in dq_only mode (round 5 on: the combine kernel's rows then count the step's kind of launch only)."""
    import csv
    f = _latest_profile("r*_k2_pmc.csv")
    if f is None:
        return None
    c = {}
    for r in csv.DictReader(open(f)):
        for key, name in (("flash", "moma::infonce_flash_kernel<512, true>"), ("combine", "moma::infonce_combine_kernel")):
            if r["kernel"].startswith(name) and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                c[(key, r["counter"])] = (float(r["mean_per_launch"]), int(r["launches"]))
    if ("flash", "FETCH_SIZE") not in c or ("flash", "WRITE_SIZE") not in c:
        return None
    import hashlib
    src = f"profiles/{os.path.basename(f)}#sha1:{hashlib.sha1(open(f, 'rb').read()).hexdigest()[:12]}"
    kb = lambda key: 2 * c[(key, "FETCH_SIZE")][0] + c[(key, "WRITE_SIZE")][0]
    whole = None
    if ("combine", "FETCH_SIZE") in c and ("combine", "WRITE_SIZE") in c and c[("combine", "FETCH_SIZE")][1] == c[("flash", "FETCH_SIZE")][1]:
        whole = int((kb("flash") + kb("combine")) * 1024)
    return int(kb("flash") * 1024), src, whole


def pmc_fracs():
    """In-kernel utilisation figures from the newest committed SQ-counter summaries (profiles/rNN_k2_pmc.csv, rNN_k1_pmc.csv;
    rocprofv3 --pmc on scripts/bench_k2.py / bench_k1.py): matrix-pipe busy share of the busy cycles, stall shares."""
    import csv
    out = {}
    for tag, pat, kernels in (("k2", "r*_k2_pmc.csv", ("moma::infonce_flash_kernel<512, true>",)),
                              ("k1", "r*_k1_pmc.csv", ("moma::k1_core_fwd_kernel", "moma::k1_core_bwd_kernel", "moma::k1_gemm_kernel",
                                                       "moma::mha_core_fwd_kernel<true>", "moma::mha_core_bwd_kernel"))):
        f = _latest_profile(pat)
        if f is None:
            continue
        vals = {}
        for r in csv.DictReader(open(f)):
            for kn in kernels:
                if r["kernel"].startswith(kn):
                    vals.setdefault(kn, {})[r["counter"]] = float(r["mean_per_launch"])
        for kn, c in vals.items():
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / SQ_WAIT_* quad-cycles (guide, cycle table)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
                key = kn.split("::")[-1]
                out[key] = {"mfma_busy_frac_of_wave_cycles": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_WAVE_CYCLES"]), 4),
                            "wait_any_frac": round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4),
                            "wait_inst_any_frac": round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4),
                            "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"), "source": os.path.basename(f)}
    return out


def k2_algorithmic(B, d, K, qbytes):
    """one queue read + q,k in + dq out + lse/loss/top1; flops = scores + P.Keys."""
    bytes_ = K * d * qbytes + 3 * B * d * 4 + B * d * 4 + 12 * B
    flops = 4.0 * B * d * (K + 1)
    return bytes_, flops


def make_opt(a, rank, world):
    return argparse.Namespace(
        distill="moma", head=a.head, feat_dim=a.feat_dim, attn="self", mem="MoCo", nce_k=a.nce_k, nce_t=0.15,
        alpha=0.999, cls=1.0, div=1.0, beta=1.0, kd_T=4.0, gpu=int(os.environ.get("LOCAL_RANK", 0)),
        multiprocessing_distributed=world > 1 or os.environ.get("MOMA_BENCH_FORCE_DIST") == "1", print_freq=a.print_freq, batch_size=a.batch_size, rank=rank,
        world_size=world, model_s=a.model, model_t=a.model_t or a.model, std_pre=None, tec_pre=None, path_t=None,
        std_strict=True, tec_strict=True, n_cls=a.n_cls, dataset="synthetic", image_size=a.image_size,
        learning_rate=a.learning_rate, momentum=0.9, weight_decay=1e-4, moma_prec=a.moma_prec, queue_dtype=a.queue_dtype,
        amp=None if a.amp == "none" else a.amp, channels_last=a.channels_last, moma_fused=True,
        shuffle_bn="per_rank", num_heads=a.num_heads, graph_student=a.graph_student, prefetch_queue=a.prefetch_queue,
        # (two processes time-slicing ONE GPU -- the CPU-side rehearsal mode -- collapse when each drives two streams)
        overlap_teacher=a.overlap_teacher and (os.environ.get("MOMA_BENCH_SAME_DEVICE") != "1" or
                                               os.environ.get("MOMA_BENCH_FORCE_OVERLAP") == "1"))


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may actually use (affinity mask and cgroup quota, not the host's total)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(a):
    """Time the CPU restatement of the same step on a bounded sample (smaller batch, few steps)."""
    from oracle.step_oracle import OracleCMO, OracleMoCo, StepOracle
    from moma_amd.backbones import model_dict
    cores = min(host_cores(), 64)
    torch.set_num_threads(cores)
    torch.manual_seed(12345)
    ms, mt = model_dict[a.model](num_classes=a.n_cls), model_dict[a.model](num_classes=a.n_cls)
    s_dim = 1280 if a.model == "effiB0" else None
    if s_dim is None:
        with torch.no_grad():
            s_dim = ms.eval()(torch.randn(2, 3, 64, 64), is_feat=True)[0][-1].shape[1]
    feat_dim = s_dim if a.head == "None" else a.feat_dim
    cmo = OracleCMO(a.head, s_dim, s_dim, feat_dim)
    contrast = OracleMoCo(feat_dim, a.nce_k, 0.15)
    run = StepOracle(ms, mt, cmo, contrast, head=a.head)
    g = torch.Generator().manual_seed(12345)
    x = torch.randn(a.cpu_batch, 3, a.image_size, a.image_size, generator=g)
    y = torch.randint(0, a.n_cls, (a.cpu_batch,), generator=g)
    run.start_epoch()
    run.step(x, y)                                   # warm-up (allocator, thread pool)
    t0 = time.time()
    for _ in range(a.cpu_steps):
        run.step(x, y)
    dt = time.time() - t0
    return {"value": round(a.cpu_batch * a.cpu_steps / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"oracle/step_oracle.py (reference op order, fp32), {a.model} pair {a.image_size}px, "
                      f"B={a.cpu_batch}, K={a.nce_k}, d={feat_dim}, {a.cpu_steps} timed steps after 1 warm-up "
                      f"({dt:.1f} s)"}


def cpu_baseline_c1(max_seconds=10.0, max_steps=60):
    """BASELINE configs[0] -- the reference's own CPU-runnable case -- beside the C2 sample: resnet8x4 student,
    resnet32x4 teacher (frozen: another architecture cannot be an EMA of the student), CIFAR-shaped 32 x 32 batches of
    8, 100 classes, `--head None` (d = s_dim = 256), K = 65536, `-c 1 -d 1 -b 1`; the step oracle (the reference's op order in
    plain fp32 torch ops), bounded by time."""
    from oracle.step_oracle import OracleCMO, OracleMoCo, StepOracle
    from moma_amd.backbones import model_dict
    cores = min(host_cores(), 64)
    torch.set_num_threads(cores)
    torch.manual_seed(12345)
    B, n_cls, K, d = 8, 100, 65536, 256
    ms, mt = model_dict["resnet8x4"](num_classes=n_cls), model_dict["resnet32x4"](num_classes=n_cls)
    run = StepOracle(ms, mt, OracleCMO("None", d, d, d), OracleMoCo(d, K, 0.15), head="None", ema=False)
    g = torch.Generator().manual_seed(12345)
    x = torch.randn(B, 3, 32, 32, generator=g)
    y = torch.randint(0, n_cls, (B,), generator=g)
    run.start_epoch()
    for _ in range(2):
        run.step(x, y)                               # warm-up (allocator, thread pool)
    t0, n = time.time(), 0
    while n < max_steps and time.time() - t0 < max_seconds:
        run.step(x, y)
        n += 1
    dt = time.time() - t0
    return {"value": round(B * n / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/step_oracle.py (reference op order, fp32), BASELINE configs[0]: resnet8x4 <- resnet32x4 (frozen), "
                      f"32px, B={B}, n_cls={n_cls}, K={K}, d={d} (--head None), {n} timed steps after 2 warm-up ({dt:.1f} s)"}


def heartbeat(period=60.0):
    """Progress line on stderr every minute (MIOpen's first-step kernel compilation can be silent for minutes)."""
    import threading
    t0 = time.time()

    def run():
        while True:
            time.sleep(period)
            log(f"... still running, {time.time() - t0:.0f} s elapsed")
    threading.Thread(target=run, daemon=True).start()


PHASES = ("started", "rccl_init", "model_built", "control_group", "wrap_self_test", "warmup_done", "timed_done", "line_printed")


_PHASE = {"name": "not started", "t0": time.time(), "log": []}


def phase(name):
    """A rank says how far it got: one stderr line per phase (a launcher -- torch.distributed.run in the driver's N > 1 run, this
    script's own parent otherwise -- relays it) and, when the parent named a file (MOMA_BENCH_PHASE_FILE), one line appended there:
    what the parent prints per rank when the job fails or runs into its deadline."""
    _PHASE["name"] = name
    _PHASE["log"].append((name, time.time() - _PHASE["t0"]))
    log(f"rank {os.environ.get('RANK', '0')}/{os.environ.get('WORLD_SIZE', '1')} phase {name} +{time.time() - _PHASE['t0']:.1f}s")
    f = os.environ.get("MOMA_BENCH_PHASE_FILE")
    if f:
        try:
            with open(f, "a") as fh:
                fh.write(f"{name} {time.time():.3f}\n")
        except OSError:
            pass


def rank_watchdog(deadline):
    """Inside every rank, launcher or not: (1) past `deadline` seconds the rank reports the phase it is stuck in and the stack of
    every thread (eight ranks parked in init_process_group or in a collective otherwise sit there until the driver's own timeout
    and leave nothing behind) and exits 124 -- the launcher then ends the job; (2) a rank that is TERMINATED (another rank failed
    first) says where it was."""
    import faulthandler
    import signal
    import threading
    faulthandler.enable()

    def where(why):
        trail = " -> ".join(f"{n} +{t:.0f}s" for n, t in _PHASE["log"]) or "not started"
        log(f"rank {os.environ.get('RANK', '0')}: {why} in phase '{_PHASE['name']}' ({trail}); stacks of all threads follow")
        faulthandler.dump_traceback(all_threads=True)

    def overdue():
        where(f"deadline of {deadline:.0f} s reached")
        os._exit(124)

    def terminated(signum, frame):
        where("terminated (SIGTERM)")
        os._exit(128 + signum)
    t = threading.Timer(deadline, overdue)
    t.daemon = True
    t.start()
    try:
        signal.signal(signal.SIGTERM, terminated)
    except ValueError:                                     # (not the main thread: a test harness)
        pass
    return t


def read_phases(path):
    try:
        rows = [l.split() for l in open(path).read().splitlines() if l.strip()]
        return [(r[0], float(r[1])) for r in rows if len(r) == 2]
    except (OSError, ValueError):
        return []


def phase_report(log_dir, n, t0):
    """per rank: the last phase reached and when (seconds after the launch)"""
    out = []
    for r in range(n):
        ph = read_phases(os.path.join(log_dir, f"bench_rank{r}.phase"))
        out.append(f"rank {r}: " + (" -> ".join(f"{name} +{t - t0:.0f}s" for name, t in ph) if ph else "no phase reached (never started?)"))
    return out


def stop_ranks(procs, grace=10.0):
    """terminate, then -- for a rank that sits in a GPU wait or a collective and ignores SIGTERM -- kill; process groups, so that
    helper processes of a rank go with it"""
    import signal
    live = [p for p in procs if p.poll() is None]
    for sig, wait in ((signal.SIGTERM, grace), (signal.SIGKILL, 5.0)):
        for p in live:
            try:
                os.killpg(p.pid, sig)
            except (ProcessLookupError, PermissionError):
                try:
                    p.send_signal(sig)
                except ProcessLookupError:
                    pass
        t_end = time.time() + wait
        while time.time() < t_end and any(p.poll() is None for p in live):
            time.sleep(0.05)
        live = [p for p in live if p.poll() is None]
        if not live:
            return


def launch_ranks(n, timeout=None, log_dir=None):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script directly (one process per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, rendezvous on 127.0.0.1) -- the reference's entry
    point spawns its own ranks too (train_student_moma.py:215-224, mp.spawn).  No torch.distributed.run in between: its launcher
    process opens the device (LaunchConfig asks torch.cuda.is_available()), one more process on the card per job.  This parent
    never touches the GPU (devices are counted from sysfs).  It relays every rank's stderr (also kept as bench_rank<r>.err in the
    log directory), prints rank 0's JSON line and returns non-zero if any rank failed or the job ran into its deadline
    (--launch_timeout; 124 then).  A failed or overdue job is stopped for good -- SIGTERM, then SIGKILL -- and leaves a report:
    per rank the phases it reached (rccl_init, control_group, wrap_self_test, warmup_done, timed_done ...) and the tail of its
    stderr.  The first N > 1 run on real hardware is one shot on a box nobody watches: what it prints is all there is."""
    import socket
    import subprocess
    import tempfile
    import threading
    if os.environ.get("MOMA_BENCH_SAME_DEVICE") != "1":
        have = visible_gpu_count()
        if have is not None and have < n:
            log(f"--gpus {n} but only {have} GPU(s) visible")
            return 2
    if timeout is None:
        timeout = float(os.environ.get("MOMA_BENCH_LAUNCH_TIMEOUT", "420"))
    if log_dir is None:
        log_dir = os.environ.get("MOMA_BENCH_LOG_DIR") or tempfile.mkdtemp(prefix="moma_bench_")
    os.makedirs(log_dir, exist_ok=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    base = dict(os.environ)
    base.pop("MOMA_BENCH_SELF_LAUNCH", None)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL's intra-node transport needs on this driver
    base.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    base.setdefault("NCCL_DEBUG", "WARN")                  # RCCL says why an init or a collective failed
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    log(f"no launcher in the environment: starting {n} ranks (deadline {timeout:.0f} s, per-rank logs in {log_dir}): {' '.join(cmd)}")
    t0 = time.time()
    procs, errs, relays = [], [], []

    def relay(r, path, stop):
        """tail rank r's stderr file into this process's stderr"""
        pos = 0
        while True:
            done = stop.is_set()
            try:
                with open(path, "r", errors="replace") as fh:
                    fh.seek(pos)
                    chunk = fh.read()
                    pos = fh.tell()
            except OSError:
                chunk = ""
            if chunk:
                sys.stderr.write(chunk)
                sys.stderr.flush()
            if done:
                return
            time.sleep(0.2)
    stop = threading.Event()
    code, out0 = 0, []
    try:
        for r in range(n):
            for f in (f"bench_rank{r}.phase", f"bench_rank{r}.err"):
                try:
                    os.remove(os.path.join(log_dir, f))
                except OSError:
                    pass
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0", MOMA_BENCH_PHASE_FILE=os.path.join(log_dir, f"bench_rank{r}.phase"))
            err = open(os.path.join(log_dir, f"bench_rank{r}.err"), "w")
            errs.append(err)
            # rank 0's stdout carries the JSON line; the other ranks' stdout joins their stderr file (progress lines stream through)
            procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else err, stderr=err, text=True,
                                          start_new_session=True))
            t = threading.Thread(target=relay, args=(r, err.name, stop), daemon=True)
            t.start()
            relays.append(t)
        reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.read().splitlines()), daemon=True)
        reader.start()
        live, why = set(range(n)), None
        while live and why is None:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0:
                    code = 128 - rc if rc < 0 else rc          # (killed by signal s: 128 + s, as a shell reports it)
                    why = f"rank {r} exited with code {rc}"
                    break
            if why is None and time.time() - t0 > timeout:
                code, why = 124, f"deadline of {timeout:.0f} s reached with rank(s) {sorted(live)} still running"
            time.sleep(0.05)
        if why is not None:
            log(f"{why}; stopping the other ranks")
            stop_ranks(procs)
        reader.join(timeout=10)
    finally:
        stop_ranks(procs, grace=2.0)                           # (also on KeyboardInterrupt / an error in this parent: no orphans)
        stop.set()
        for t in relays:
            t.join(timeout=5)
        for e in errs:
            e.close()
    lines = [l for l in out0 if l.startswith("{")]
    for l in out0:
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if code != 0 or not lines:
        log(f"ranks exited with code {code}{'' if lines else ' and no JSON line'}; how far every rank got:")
        for l in phase_report(log_dir, n, t0):
            log("  " + l)
        for r in range(n):
            try:
                tail = open(os.path.join(log_dir, f"bench_rank{r}.err"), errors="replace").read().splitlines()[-8:]
            except OSError:
                tail = []
            for l in tail:
                log(f"  [rank {r} stderr] {l}")
        return code or 1
    print(lines[-1], flush=True)
    return 0


def main():
    a = parse()
    # (MOMA_BENCH_SELF_LAUNCH=1: also --gpus 1 goes through the parent path -- how the one-GPU box rehearses the exact code of --gpus N)
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or os.environ.get("MOMA_BENCH_SELF_LAUNCH") == "1"):
        raise SystemExit(launch_ranks(a.gpus, timeout=a.launch_timeout))       # before any GPU call in this process
    heartbeat()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 or os.environ.get("MOMA_BENCH_FORCE_DIST") == "1":
        rank_watchdog(float(os.environ.get("MOMA_BENCH_RANK_DEADLINE", "480")))
        phase("started")
    if world != a.gpus and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MoMA hot path is a HIP library (no CPU fallback)")
    # rehearsal knobs (never set by the driver): run N ranks on ONE GPU with gloo to exercise the N>1 code path
    if os.environ.get("MOMA_BENCH_SAME_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # MOMA_BENCH_FORCE_DIST=1 (rehearsal, never set by the driver): the N>1 code path -- RCCL communicator and its watchdog thread,
    # DDP reducer, hook-launched criterion all-reduce next to the captured teacher graphs -- with a ONE-rank group on one GPU
    distributed = world > 1 or os.environ.get("MOMA_BENCH_FORCE_DIST") == "1"
    if distributed:
        import datetime
        os.environ.setdefault("NCCL_DEBUG", "WARN")               # (also under a launcher that is not ours: RCCL says why it failed)
        # a collective that cannot complete raises after this long instead of after torch's default ten minutes -- inside the
        # rank's own deadline (rank_watchdog), so that the error, not the watchdog, is what the log ends with
        dist.init_process_group(os.environ.get("MOMA_BENCH_BACKEND", "nccl"),     # "nccl" = RCCL over xGMI
                                timeout=datetime.timedelta(seconds=float(os.environ.get("MOMA_BENCH_COLLECTIVE_TIMEOUT", "300"))))
        if dist.get_backend() == "nccl":
            # first contact with the communicator HERE, not inside the model wrap: RCCL builds its rings lazily at the first collective
            t = torch.ones(1, device=dev)
            dist.all_reduce(t)
            torch.cuda.synchronize()
            assert int(t.item()) == world, f"all-reduce over {world} ranks returned {t.item()}"
        phase("rccl_init")
    torch.backends.cudnn.benchmark = bool(a.miopen_find)

    from moma_amd import ops
    from moma_amd.train_student_moma import build_training
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    from moma_amd.helper.loops_moma import train_distill_moma
    from moma_amd.dataset.synthetic import SyntheticLoader

    opt = make_opt(a, rank, world)
    torch.manual_seed(12345)                         # identical initial weights on every rank
    model_s, model_t, module_list, criterion_list, _tr, contrast, optimizer = build_training(opt, dev)
    trainer = ContrastTrainer(opt)
    trainer.grad_sync_single_rank = distributed and world == 1      # (the rehearsal runs the hook-launched all-reduce on its one rank)
    if opt.amp == "fp16":
        opt._grad_scaler = torch.amp.GradScaler("cuda")       # as train_student_moma.main_worker does
    if distributed:
        from moma_amd.learning.ddp import wrap_student, control_group
        from moma_amd.learning.ddp import broadcast_module_state
        phase("model_built")
        control_group()                                            # (the host-side gloo group of the flat wrap: created here, by every rank)
        phase("control_group")
        ddp_s = wrap_student(model_s, device_ids=[local])          # MOMA_DP=ddp keeps the stock reducer
        broadcast_module_state([criterion_list[2], model_t])       # (under no wrap: identical by seed in the reference)
        phase("wrap_self_test")
        opt.gpu = local
        module_list = [ddp_s] + list(module_list)[1:]
    rec = EventRecorder()
    ops.set_event_recorder(rec)
    kev = KernelEvents()
    ops.set_kernel_event_provider(kev)

    loader_w = SyntheticLoader(a.warmup, a.batch_size, a.image_size, a.n_cls, 12345 + rank, dev)
    loader_t = SyntheticLoader(a.steps, a.batch_size, a.image_size, a.n_cls, 12345 + rank, dev)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    log(f"rank {rank}/{world}: model + queue built; warm-up {a.warmup} steps (first step JIT-compiles MIOpen kernels)")
    import contextlib
    quiet = contextlib.redirect_stdout(sys.stderr)      # stdout carries exactly one JSON line
    amp_ctx = lambda: torch.autocast("cuda", dtype={"bf16": torch.bfloat16, "fp16": torch.float16}.get(opt.amp), enabled=opt.amp is not None)
    # ---- untimed: whatever would otherwise run for the FIRST time inside the timed region (round 2's driver line lost 7 % to one
    # ~65 ms first step).  (1) Every epoch opens with the teacher in eval mode for its first forward (reference
    # helper/loops_moma.py:227): a (shape, eval) variant the warm-up epoch serves once, eagerly -- ~700 launches of host time.
    # It is served here until its HIP graph exists (eval-mode BatchNorm: no state changes), on the stream the loop replays on,
    # and BEFORE the warm-up steps: a graph capture empties the caching allocator's pool (torch.cuda.graph does), which the
    # warm-up steps then grow back -- captured after them, the first timed step would pay the hipMallocs (seen once: 350 ms).
    if getattr(opt, "graph_teacher", True) and a.warmup > 0:
        from moma_amd.helper.graphs import GraphedInference
        teacher = trainer._graphed_teacher = GraphedInference(model_t)
        if opt.overlap_teacher:
            trainer._side_stream = torch.cuda.Stream(device=dev)
        side = getattr(trainer, "_side_stream", None)
        x0 = next(iter(loader_w))[0]
        model_t.eval()
        with torch.cuda.stream(side if side is not None else torch.cuda.current_stream()), amp_ctx():
            primed = teacher.prime(x0, is_feat=True)
        torch.cuda.synchronize()
        log(f"teacher eval-mode variant graphed before the warm-up: {primed}")
    if a.warmup > 0:
        with quiet:
            train_distill_moma(0, loader_w, module_list, criterion_list, trainer, contrast, optimizer, opt)
    # (2) the instrumented paths (HIP event creation, the first dispatch that carries events) on calls without side effects
    rec.enabled = True
    kev.enabled = True
    kd = criterion_list[2]
    with torch.no_grad():
        d_feat = contrast.memory.shape[1]
        qd = torch.nn.functional.normalize(torch.randn(a.batch_size, d_feat, device=dev))
        if hasattr(kd, "atts_k"):
            kd.atts_k(qd)
        mem = contrast._bf16_shadow() if (contrast.memory.dtype == torch.float32 and a.moma_prec == "bf16") else contrast.memory
        ops.infonce_fused(qd, qd, mem, contrast.T, a.moma_prec)          # forward only: no enqueue, no state change
        del qd
    torch.cuda.synchronize()
    rec.events.clear()
    kev.pairs.clear()
    barrier()
    if distributed:
        phase("warmup_done")
    log(f"warm-up done; timing {a.steps} steps")
    opt.step_events = []
    sg = getattr(trainer, "_step_graphs", None)
    replays0 = sg.replays if sg is not None else 0
    e_start = torch.cuda.Event(enable_timing=True)
    t0, c0 = time.perf_counter(), time.thread_time()
    e_start.record()
    with quiet:
        _acc, loss_avg = train_distill_moma(1, loader_t, module_list, criterion_list, trainer, contrast, optimizer, opt)
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        phase("timed_done")
    sg = getattr(trainer, "_step_graphs", None)
    replayed = (sg.replays if sg is not None else 0) - replays0
    rec.enabled = False
    kev.enabled = False
    # per-step times: GPU = between the HIP events recorded at the end of consecutive steps (stream time, includes queueing
    # behind the previous step); host = when the host finished issuing the step
    # (host issue is WALL time: the graph-served loop paces itself two steps ahead of the GPU (helper/step_graph.py:_throttle), so
    #  its median tends to the GPU's step time.  host cpu = CPU time of the issuing thread: with the pacing -- a poll of the step's
    #  event between short sleeps -- it is what a step costs the host, ~7 ms.  Every wait INSIDE this runtime spins, blocking-sync
    #  events included (scripts/diag_blocking_event.py): without the pacing the thread burned wall = CPU time in hipGraphLaunch,
    #  44.5 ms per 39.8 ms step in round 4's driver line.  host_issue_ms_min = a step issued into an empty queue.)
    step_gpu, step_host, step_cpu, prev_e, prev_t, prev_c = [], [], [], e_start, t0, c0
    for t_host, ev, t_cpu in opt.step_events:
        step_gpu.append(prev_e.elapsed_time(ev))
        step_host.append((t_host - prev_t) * 1e3)
        step_cpu.append((t_cpu - prev_c) * 1e3)
        prev_e, prev_t, prev_c = ev, t_host, t_cpu
    opt.step_events = None
    if rank == 0 and step_gpu:
        log("per-step ms (gpu | host issue | host cpu): " + " ".join(f"{g:.1f}|{h:.1f}|{c:.1f}" for g, h, c in zip(step_gpu, step_host, step_cpu)))

    # a timed region whose loss is not finite measured a broken step (round 3 saw one produce a normal-looking line): no line
    import math
    bad = torch.tensor([0.0 if math.isfinite(loss_avg) else 1.0], device=dev)
    if distributed:
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    if bad.item() != 0.0:
        log(f"rank {rank}: mean loss of the timed steps is {loss_avg} -- a non-finite loss on some rank: no result line")
        if distributed:
            dist.destroy_process_group()
        raise SystemExit(3)
    med = lambda v: sorted(v)[len(v) // 2] if v else 0.0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    spread = None
    per_rank = None
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # per-rank view of the timed region: median step (GPU / host issue), the step's collective and the buffer broadcast
        mine = torch.tensor([med(step_gpu), med(step_host), rec.mean_ms("dp_allreduce_grads") or 0.0,
                             rec.mean_ms("dp_buffer_broadcast") or 0.0, loss_avg, med(step_cpu)], device=dev, dtype=torch.float64)
        rows = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, mine)
        per_rank = [[round(float(v), 3) for v in r.tolist()] for r in rows]
        # replicas must still be bit-identical: student (DDP all-reduce), trainable criterion modules (the hook-launched flat
        # all-reduce), EMA teacher (updated locally from identical student weights)
        def csum(mod):
            return torch.stack([p.detach().double().sum() for p in mod.parameters()]).sum()
        mine = torch.stack([csum(model_s), csum(criterion_list[2]), csum(model_t)])
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        spread = [float(v) for v in (hi - lo).tolist()]
    dt = float(t.item())

    if rank == 0:
        d = contrast.memory.shape[1]
        qbytes = contrast.memory.element_size()
        k2_ms = kev.mean_ms()                         # the one-pass kernel alone (events recorded by the library)
        k2_call_ms = rec.mean_ms("moma_infonce_fused")  # whole C-ABI call between two HIP events recorded AROUND it
        k2_kernels_ms = kev.mean_ms(2)                # first kernel's start .. last kernel's end, on the dispatches
        bytes_, flops = k2_algorithmic(a.batch_size, d, a.nce_k, qbytes)
        # governing bound = the larger ideal time: HBM for a 4-byte queue, MFMA for bf16 at B=256
        t_hbm, t_mfma = bytes_ / (HBM_PEAK_GBS * 1e9), flops / (MFMA_BF16_PEAK_TFLOPS * 1e12)
        if a.moma_prec == "fp32":
            # exact-fp32 policy: the f32-input MFMA (an exact fp32 fma chain) runs at the fp32 vector rate, 1/16 of the bf16 rate --
            # that rate, not HBM, bounds the one-pass kernel (34.4 GFLOP -> 219 us at d = 512 against 17 us for its 135 MB)
            ach = flops / (k2_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None}
        elif a.moma_prec == "bf16" and t_mfma > t_hbm:
            ach = flops / (k2_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None}
        else:
            ach = bytes_ / (k2_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None}
        if (a.batch_size, d, a.nce_k, a.queue_dtype) == (256, 512, 65536, "bf16"):
            tr = k2_pmc_traffic()                       # bytes per launch: PMC passes of the same kernel, committed summary
            if tr is not None:
                roof["traffic"], roof["traffic_source"] = tr[0], tr[1] + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/bench_k2.py; not measured in this run)"
                # every kernel of the K2 call: the one-pass kernel + the combine that reads its split-K partials back (the side-stream
                # sweep that pre-fetched the queue -- a second 67 MB read per step -- is off since round 5: opt.prefetch_queue)
                roof["traffic_whole_call"] = tr[2]
                roof["traffic_whole_call_note"] = ("one-pass kernel + combine, same PMC summary; queue pre-fetch sweep "
                                                   + ("ON (its 67 MB per step are NOT in this figure)" if a.prefetch_queue else "off"))
        k1f, k1b, k4 = rec.mean_ms("moma_mha_fwd"), rec.mean_ms("moma_mha_bwd"), rec.mean_ms("moma_ema_multi")
        k1g = rec.mean_ms("moma_mha_fwd_group2")
        n_par = sum(p.numel() for p in model_s.parameters())
        d_att = d
        other = {"pmc": pmc_fracs()}
        if k4:
            # (two K4 calls per step: backbone pair and mlp head pair; the mean is dominated by the 48 MB backbone call)
            other["k4_ema"] = {"ms_per_call_in_step": round(k4, 4), "algorithmic_bytes_backbone": 12 * n_par,
                               "note": "HIP events around the C-ABI call on the side stream, concurrent with the student forward; "
                                       "alone (scripts/bench_k4.py, profiles/) the 48.2 MB call runs in ~10 us"}
        if k1f:
            fl = 8.0 * a.batch_size * d_att * d_att + 4.0 * a.batch_size * a.batch_size * d_att
            other["k1_attention"] = {"fwd_ms_per_module_call": round(k1f, 4), "bwd_ms_per_call": round(k1b, 4) if k1b else None,
                                     "fwd_ms_two_modules_grouped": round(k1g, 4) if k1g else None,
                                     "fwd_flops": fl, "fwd_mfma_frac_wallclock": round(fl / (k1f * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 5),
                                     "note": "launch / latency bound at N = batch (0.67 GFLOP per module call); HIP events around the C-ABI "
                                             "call (3 launches each; atts_k + atts_queue run as ONE group of 3 launches on the side stream); "
                                             "in-kernel matrix-pipe shares are in pmc.k1_*"}
        if roof["bound"] == "mfma" and a.moma_prec == "bf16" and d <= 512:
            # context, not the contract's figure: the peak above is the 2.4 GHz one; inside this kernel's loop the chip holds
            # 1.80 GHz (s_memtime / s_memrealtime stamps of the diagnostic build, profiles/r04_flash_kernel_timeline.txt)
            roof["sustained_clock_ghz_in_kernel"] = 1.80
            roof["frac_of_peak_at_sustained_clock"] = round(roof["frac"] * 2.4 / 1.80, 4)
        roof["other"] = other
        kname = ("infonce_f32_flash_kernel (K2 one pass over the fp32 queue on the f32-input MFMA; moma_infonce_fused)"
                 if (a.moma_prec == "fp32" and a.queue_dtype == "fp32" and d in (128, 256, 384, 512, 768, 1024, 1280)) else
                 "staged exact-fp32 K2 (logits -> row reduction -> gradient GEMM)" if a.moma_prec == "fp32" else
                 ("infonce_small_kernel (K2 one pass over the queue, key-half split for B <= 64; moma_infonce_fused)" if a.batch_size <= 64 else
                  "infonce_flash_kernel (K2 one pass over the queue; moma_infonce_fused)") if d <= 512 else
                 "infonce_wide_scores_kernel + infonce_wide_pv2_kernel (K2 over a wide queue, d > 512: the two passes over the queue; "
                 "like the one-pass line the Q pre-pack in front and the combine behind are in whole_call_ms only)")
        roof.update({"kernel": kname,
                     "ms_per_launch": round(k2_ms, 4), "whole_call_ms": round(k2_kernels_ms, 4),
                     "whole_call_ms_between_host_events": round(k2_call_ms, 4),
                     "whole_call_note": "whole_call_ms = start of the call's first kernel .. end of its last kernel (events on the "
                                        "dispatches; the query arrives packed from K1, so the call is the one-pass kernel + the combine); "
                                        "_between_host_events adds the latency of two event packets recorded around the C-ABI call",
                     "algorithmic_bytes": bytes_, "algorithmic_flops": flops,
                     "hbm_frac": round(bytes_ / (k2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "mfma_frac": round(flops / (k2_ms * 1e-3) / 1e12 / (MFMA_F32_PEAK_TFLOPS if a.moma_prec == "fp32" else MFMA_BF16_PEAK_TFLOPS), 4),
                     "other_ms": {k: round(rec.mean_ms(k), 4) for k in rec.events if k != "moma_infonce_fused"}})
        out = {
            "metric": "images/sec MoMA train step (EffNet-B0 224px, K=65536)",
            "value": round(world * a.batch_size * a.steps / dt, 2),
            "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "ms_per_step_median": round(sorted(step_gpu)[len(step_gpu) // 2], 3) if step_gpu else None,
            "ms_per_step_max": round(max(step_gpu), 3) if step_gpu else None,
            "ms_first_step": round(step_gpu[0], 3) if step_gpu else None,
            "host_issue_ms_median": round(med(step_host), 3) if step_host else None,
            "host_issue_ms_min": round(min(step_host), 3) if step_host else None,
            "host_cpu_ms_median": round(med(step_cpu), 3) if step_cpu else None,
            "host_note": "host_cpu_ms_median = CPU time of the issuing thread per step (the graph-served loop sleeps between polls of "
                         "the step-before-last's event instead of spinning inside the runtime); host_issue_ms_median = wall time, "
                         "paced by the GPU; host_issue_ms_min = a step issued into an empty queue",
            "loss_mean_timed_steps": round(loss_avg, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.moma_prec == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{2 if a.model.startswith('vit_small') else 1}]: {a.model} student+teacher" if (a.model_t or a.model) == a.model else
                                    f"BASELINE configs[4], single-GPU form: {a.model} student <- {a.model_t} teacher (frozen: architectures differ)") +
                                   f" (random init), synthetic "
                                   f"{a.image_size}x{a.image_size} RGB, per-GPU batch {a.batch_size}, queue K={a.nce_k} "
                                   f"x d={d} ({a.queue_dtype}), head={a.head}, attn=self ({a.num_heads} heads), -c 1 -d 1 -b 1, "
                                   f"alpha=0.999, T=0.15, SGD; backbones autocast={a.amp} (BN+SiLU / depthwise / SE on the library's helper "
                                   f"kernels: MOMA_BN={os.environ.get('MOMA_BN', 'hip')} MOMA_DW={os.environ.get('MOMA_DW', 'hip')} "
                                   f"MOMA_SE={os.environ.get('MOMA_SE', 'hip')}), KD kernels {a.moma_prec}",
                       "global_batch": world * a.batch_size, "parallelism": f"dp{world}", "queue": "per-rank",
                       "step_graphs": {"enabled": bool(a.graph_student), "timed_steps_replayed": int(replayed),
                                       "DEBUG_CLR_GRAPH_PACKET_CAPTURE": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"),
                                       "host_pacing": os.environ.get("MOMA_GRAPH_THROTTLE", "1") == "1",
                                       "HSA_KERNARG_POOL_SIZE": os.environ.get("HSA_KERNARG_POOL_SIZE")},
                       "queue_prefetch_sweep": bool(a.prefetch_queue)},
            "roofline": roof,
        }
        if distributed:      # what the N>1 line was measured with (the driver checks it against its own launch)
            cols = list(zip(*per_rank))
            out["dist"] = {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                           "per_rank_ms_per_step_median": list(cols[0]), "per_rank_host_issue_ms_median": list(cols[1]),
                           "per_rank_allreduce_grads_ms": list(cols[2]), "per_rank_buffer_broadcast_ms": list(cols[3]),
                           "per_rank_loss": list(cols[4]), "per_rank_host_cpu_ms_median": list(cols[5]),
                           "timing_note": "HIP events on the rank's main stream around FlatDataParallel.allreduce_grads (cat + the "
                                          "collective + copy back; includes waiting for the slowest rank to arrive) and around the "
                                          "flat buffer broadcast in front of the student forward; host issue = wall time the host "
                                          "spent issuing a step; 0.0 = that path did not run (MOMA_DP=ddp: the reducer's buckets)",
                           "replica_checksum_spread": {"student": spread[0], "criterion": spread[1], "ema_teacher": spread[2]},
                           "criterion_allreduce_launches": int(getattr(trainer, "grad_sync_launches", 0)),
                           "overlap_teacher": bool(opt.overlap_teacher), "graph_teacher": bool(getattr(opt, "graph_teacher", True)),
                           "dp_wrap": type(ddp_s).__name__,
                           "collective": ("ONE flat gradient all-reduce per step (student + trainable criterion modules) behind the "
                                          "backward + one flat buffer broadcast per forward; per-rank queue, no data-path gather")
                           if type(ddp_s).__name__ == "FlatDataParallel" else
                           ("DDP bucketed gradient all-reduce (student) + one flat async all-reduce of the "
                            "trainable criterion modules per step; per-rank queue, no data-path gather")}
        log(f"timed region: {dt:.2f} s; {out['value']} images/sec")
        if world == 1 and not a.no_cpu_baseline:
            log("timing the CPU restatement (bounded sample) ...")
            out["cpu_baseline"] = cpu_baseline(a)
            out["cpu_baseline_c1"] = cpu_baseline_c1()
        print(json.dumps(out), flush=True)
    if distributed:
        phase("line_printed")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
