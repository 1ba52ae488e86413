set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04f_gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc $rc"; tail -8 gpurun_out/r04f_gpu_tests.log
