"""N `--reproducible` CLI runs per variant: are the checkpoints bit-identical, and if not, which flag makes them so?
usage: python scripts/diag_cli_repro.py [N]      (checkpoints go to a temporary directory)"""
import os, subprocess, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
BASE = [sys.executable, os.path.join(ROOT, "train_student_moma.py"), "--distill", "moma", "--model_s", "resnet8x4", "--model_t", "resnet8x4",
        "--dataset", "cifar100", "--n_cls", "2", "--batch_size", "32", "--epochs", "2", "--steps_per_epoch", "7", "--nce_k", "1024",
        "--head", "mlp", "--feat_dim", "128", "-c", "1", "-d", "1", "-b", "1", "--reproducible"]


def flat(x, path, out):
    if isinstance(x, dict):
        for k in x:
            flat(x[k], f"{path}.{k}", out)
    elif isinstance(x, (list, tuple)):
        for i, v in enumerate(x):
            flat(v, f"{path}[{i}]", out)
    elif torch.is_tensor(x):
        out[path] = x


for label, extra in (("graphs on (default)", []), ("--no_graph_student", ["--no_graph_student"]), ("--no_graph_teacher", ["--no_graph_teacher"]),
                     ("--no_overlap_teacher", ["--no_overlap_teacher"]), ("both graphs off", ["--no_graph_student", "--no_graph_teacher"])):
    states = []
    with tempfile.TemporaryDirectory() as tmp:
        for r in range(N):
            p = subprocess.run(BASE + extra + ["--save_root", os.path.join(tmp, str(r))], capture_output=True, text=True, timeout=900, cwd=ROOT)
            if p.returncode != 0:
                print(label, "FAILED", p.stderr[-600:], flush=True)
                break
            ck = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(tmp, str(r))) for f in fs if f == "ckpt_last.pth"][0]
            out = {}
            flat(torch.load(ck, map_location="cpu", weights_only=False), "ckpt", out)
            states.append(out)
    if len(states) < N:
        continue
    worst, where, ndiff = 0.0, None, 0
    for s in states[1:]:
        bad = [k for k in states[0] if not torch.equal(states[0][k], s[k])]
        ndiff += bool(bad)
        for k in bad:
            dlt = (states[0][k].double() - s[k].double()).abs().max().item()
            if dlt > worst:
                worst, where = dlt, k
    print(f"{label}: {N} runs, {ndiff} differ from the first; largest difference {worst:.3e} in {where}", flush=True)
