set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --deselect tests/test_gpu_kernels.py -k "not cli" > gpurun_out/r2_full2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_full2.log
tail -30 gpurun_out/r2_full2.log
python -m pytest tests/test_gpu_cli.py -q -m gpu -x > gpurun_out/r2_cli.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_cli.log
tail -12 gpurun_out/r2_cli.log
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "effnet or graphed or bn_ or dwconv or se_gate" > gpurun_out/r2_bb.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_bb.log
tail -3 gpurun_out/r2_bb.log
