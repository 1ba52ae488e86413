set -o pipefail
mkdir -p gpurun_out
SOL_VERBOSE=1 timeout -k 10 600 python tests/tools/split_sweep.py c1x c2 c3 c5 c3h c3hi --budgets -1 --slacks 3 --reinsert 8 2>&1 | grep -E "probe|split  -1" | tee gpurun_out/r04l_radius_probe.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04l_gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -5 gpurun_out/r04l_gpu_tests.log
