set -o pipefail
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -w tests/tools/micro/valu_rate.hip -o gpurun_out/valu_rate && timeout -k 10 120 gpurun_out/valu_rate > gpurun_out/r02a_valu_rate.log 2>&1
cat gpurun_out/r02a_valu_rate.log
timeout -k 10 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r02a_bench_tests.log 2>&1; echo "bench tests rc $?"; tail -15 gpurun_out/r02a_bench_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size or c4" > gpurun_out/r02a_fullsize.log 2>&1; echo "fullsize rc $?"; tail -5 gpurun_out/r02a_fullsize.log
timeout -k 10 600 python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err; echo "bench rc $?"; cat gpurun_out/r02a_bench.json; tail -5 gpurun_out/r02a_bench.err
