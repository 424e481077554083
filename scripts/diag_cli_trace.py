"""One CLI run (`--reproducible`) in this process with the per-step trace kept; writes the losses bit-exactly (hex) to argv[1].
Driver mode (no arguments besides N): starts N such processes per variant and reports the first step at which their traces part.
usage: python scripts/diag_cli_trace.py N          |  (internal) python scripts/diag_cli_trace.py --one <out file> [CLI args]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BASE = ["--distill", "moma", "--model_s", "resnet8x4", "--model_t", "resnet8x4", "--dataset", "cifar100", "--n_cls", "2", "--batch_size", "32",
        "--epochs", "2", "--steps_per_epoch", "7", "--nce_k", "1024", "--head", "mlp", "--feat_dim", "128", "-c", "1", "-d", "1", "-b", "1",
        "--reproducible"]

if len(sys.argv) > 2 and sys.argv[1] == "--one":
    out, args = sys.argv[2], sys.argv[3:]
    from moma_amd import train_student_moma as T
    opt = T.parse_option(args)
    opt.trace = []
    import hashlib
    import torch
    grads = []                                     # per optimizer step: (shape, digest of the gradient) of every parameter, in order
    _step = torch.optim.SGD.step

    def step(self, *a, **kw):
        if len(grads) < 3:
            grads.append([(tuple(p.shape), "none" if p.grad is None else hashlib.sha256(p.grad.detach().float().cpu().numpy().tobytes()).hexdigest()[:12])
                          for g in self.param_groups for p in g["params"]])
        return _step(self, *a, **kw)
    torch.optim.SGD.step = step
    T.main_worker(0, 1, opt)
    torch.cuda.synchronize()
    with open(out, "w") as f:
        for loss, index, loss_kd in opt.trace:
            f.write(f"{float(loss).hex()} {index} {float(loss_kd).hex()}\n")
    with open(out + ".grads", "w") as f:
        for i, g in enumerate(grads):
            for j, (shape, dg) in enumerate(g):
                f.write(f"{i} {j} {shape} {dg}\n")
    sys.exit(0)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for label, extra in (("graphs on", []), ("--no_graph_teacher", ["--no_graph_teacher"]), ("--no_graph_student", ["--no_graph_student"])):
    traces, gradlogs = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for r in range(N):
            out = os.path.join(tmp, f"t{r}.txt")
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", out] + BASE + extra + ["--save_root", os.path.join(tmp, f"s{r}")],
                               capture_output=True, text=True, timeout=900, cwd=ROOT)
            if p.returncode != 0:
                print(label, "FAILED", p.stderr[-800:], flush=True)
                break
            traces.append(open(out).read().splitlines())
            gradlogs.append(open(out + ".grads").read().splitlines())
    if len(traces) < N:
        continue
    firsts = []
    for t in traces[1:]:
        firsts.append(next((i for i, (a, b) in enumerate(zip(traces[0], t)) if a != b), None))
    print(f"{label}: {N} processes, {len(set(map(tuple, traces)))} distinct traces; first differing step vs run 0: {firsts}", flush=True)
    # the first gradient (optimizer step, parameter index, shape) that is not the same in every process
    for row in zip(*gradlogs):
        if len(set(row)) > 1:
            step_i, j, rest = row[0].split(" ", 2)
            n_params = sum(1 for l in gradlogs[0] if l.startswith("0 "))
            differing = sorted({r.split(" ", 2)[1] for rows in zip(*gradlogs) if len(set(rows)) > 1 and rows[0].startswith(step_i + " ") for r in rows[:1]}, key=int)
            print(f"    first differing gradient: optimizer step {step_i}, parameter {j} of {n_params}, {rest.rsplit(' ', 1)[0]}; parameters that differ in that step: {differing[:40]}", flush=True)
            break
    else:
        print("    gradients of the first three steps: identical in every process", flush=True)
