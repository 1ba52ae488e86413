set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_world_tree.py -x -q -m gpu > gpurun_out/r04c_tree_tests.log 2>&1; rc=$?; echo "tree tests rc $rc"; tail -5 gpurun_out/r04c_tree_tests.log
[ $rc -eq 0 ] || exit $rc
SOL_VERBOSE=1 timeout -k 10 300 python tests/tools/perf_quick.py c1 c2 c3 test --spp 64 2>&1 | grep -v "work order" | tee gpurun_out/r04c_perf.txt
SOL_VERBOSE=1 timeout -k 10 300 python tests/tools/split_sweep.py c3h c5 --budgets 30 --slacks 3 2>&1 | grep -E "pre-split|split " | tee -a gpurun_out/r04c_perf.txt
timeout -k 10 300 python tests/tools/phase_counts.py test c3 c3h c5 2>&1 | tee gpurun_out/r04c_phase_counts.txt
