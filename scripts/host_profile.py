"""Host-side cost of the training step: cProfile over a few bench steps (GPU-bound runs hide it; the multi-GPU run shares the
host between 8 ranks).  usage: python scripts/host_profile.py [steps]   -> prints the top cumulative / self entries"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = sys.argv[1] if len(sys.argv) > 1 else "12"
sys.argv = ["bench.py", "--steps", steps, "--warmup", "6", "--no_cpu_baseline"]
import bench
pr = cProfile.Profile()
import moma_amd.helper.loops_moma as L
orig = L.train_distill_moma
state = {"n": 0}
def wrapped(*a, **k):
    state["n"] += 1
    if state["n"] >= 2:                 # the timed epoch (the warm-up epoch runs unprofiled)
        pr.enable()
        try:
            return orig(*a, **k)
        finally:
            pr.disable()
    return orig(*a, **k)
L.train_distill_moma = wrapped
bench.main()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000], file=sys.stderr)
