set -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/round_gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -5 gpurun_out/round_gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/round_bench.json 2> gpurun_out/round_bench.err; echo "bench rc $?"; cat gpurun_out/round_bench.json | cut -c1-1500
