set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_world_tree.py -x -q -m gpu -k "needle or heterogeneous or variants_are_bit_identical or device_built_tree" > gpurun_out/r04s_needle_tests.log 2>&1; echo "needle tests rc $?"; tail -6 gpurun_out/r04s_needle_tests.log
(cd tests/tools && timeout -k 10 300 python gpu_full_oracle.py c3h 16 1920 1080 default) 2>&1 | tail -2
timeout -k 10 300 python tests/tools/perf_quick.py c3h --spp 64 --phases
SOL_SPLIT=0 timeout -k 10 300 python tests/tools/perf_quick.py c3h --spp 64 --phases
