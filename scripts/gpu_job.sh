set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 6 --no_cpu_baseline > gpurun_out/r2_bench4.json 2> gpurun_out/r2_bench4.err; tail -2 gpurun_out/r2_bench4.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r2_bench4.json")); r=j["roofline"]
print("value",j["value"],"ms/step",j["ms_per_step"],"k2 ms",r["ms_per_launch"],"frac",r["frac"],"call",r["whole_call_ms"],r["other_ms"])
PY
python -m pytest tests/test_gpu_step.py tests/test_gpu_cli.py -q -m gpu -x > gpurun_out/r2_step5.log 2>&1; echo "rc=$?" >> gpurun_out/r2_step5.log; tail -3 gpurun_out/r2_step5.log
