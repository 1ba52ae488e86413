set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04a_gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc $rc"; tail -5 gpurun_out/r04a_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/variants.py run --spp 64 --scenes "c1 c2 c3 test" r3 default r3 default > gpurun_out/r04a_medium_ab.txt 2>&1; echo "variants rc $?"; cat gpurun_out/r04a_medium_ab.txt
timeout -k 10 300 python bench.py --workload profiling --no-cpu-baseline --no-build > gpurun_out/r04a_profiling_bench.json 2> gpurun_out/r04a_profiling_bench.err; echo "bench rc $?"; cut -c1-3000 gpurun_out/r04a_profiling_bench.json
