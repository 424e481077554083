"""The distributed bring-up the driver's multi-GPU run goes through, rehearsed on CPU: `python -m torch.distributed.run`
with 2 ranks (gloo, 127.0.0.1) -> the trainer's `main()` torchrun branch -> `main_worker` ->
`BaseTrainer.init_ddp_environment` -> DDP(student) -> one epoch -> checkpoints (shared + one per rank)."""
import json
import os
import socket
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_main_torchrun_branch_two_ranks_gloo(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_torchrun_cpu_worker.py"), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    seen = [json.load(open(tmp_path / f"seen_rank{i}.json")) for i in range(2)]
    for i, s in enumerate(seen):
        assert s["initialized"] and s["backend"] == "gloo"
        assert s["rank"] == i and s["local_rank"] == i and s["node_rank"] == 0        # global rank = node * n_local + local
        assert s["world_size"] == 2 and s["ngpus_per_node"] == 2 and s["local_group_size"] == 2
    assert "images/sec" in r.stdout and "best accuracy" in r.stdout
    files = [f for _, _, fs in os.walk(tmp_path) for f in fs]
    assert "ckpt_last.pth" in files and "ckpt_last_rank0.pth" in files and "ckpt_last_rank1.pth" in files
    ck = {f: os.path.join(dp, f) for dp, _, fs in os.walk(tmp_path) for f in fs if f.startswith("ckpt_last")}
    q0 = torch.load(ck["ckpt_last_rank0.pth"], weights_only=False)["contrast"]
    q1 = torch.load(ck["ckpt_last_rank1.pth"], weights_only=False)["contrast"]
    assert q0["_extra_state"]["index"] == q1["_extra_state"]["index"] == 12           # 3 steps x B = 4, per-rank queue
    assert not torch.equal(q0["memory"], q1["memory"])                                # different data shard per rank
    shared = torch.load(ck["ckpt_last.pth"], weights_only=False)
    assert shared["epoch"] == 1 and "model_t" in shared and "criterion_kd" in shared


    # ---- resume: rank 1's per-rank file is made to look like another epoch's (a job killed between the two writes / stale
    # files in the folder): it must be ignored -- queue from the shared file, no RNG restore -- while rank 0's is used
    rs = torch.load(ck["ckpt_last_rank1.pth"], weights_only=False)
    rs["epoch"] = 7
    torch.save(rs, ck["ckpt_last_rank1.pth"])
    cmd2 = cmd[:-1] + [str(tmp_path), "--resume", ck["ckpt_last.pth"], "--epochs", "2"]
    cmd2[cmd2.index("--master-port") + 1] = str(_free_port())
    r2 = subprocess.run(cmd2, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    assert "is from epoch 7 but the shared checkpoint from epoch 1: ignored" in r2.stdout
    assert "per-rank state found" in r2.stdout and "per-rank state absent" in r2.stdout
    again = torch.load(ck["ckpt_last_rank1.pth"], weights_only=False)
    assert again["epoch"] == 2 and again["contrast"]["_extra_state"]["index"] == 24      # continued from the shared pointer (12)
    assert not any(f.endswith(".pth") is False and ".tmp" in f for _, _, fs in os.walk(tmp_path) for f in fs)   # no leftovers
