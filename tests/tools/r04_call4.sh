set -o pipefail
mkdir -p gpurun_out
SOL_REINSERT=6 timeout -k 10 600 python -m pytest tests/test_world_tree.py -x -q -m gpu > gpurun_out/r04d_tree_tests.log 2>&1; rc=$?; echo "tree tests (6 reinsertion rounds) rc $rc"; tail -5 gpurun_out/r04d_tree_tests.log
[ $rc -eq 0 ] || exit $rc
SOL_VERBOSE=1 SOL_REINSERT=8 timeout -k 10 200 python tests/tools/perf_quick.py c3 --spp 16 2>&1 | grep -v "work order" | tee gpurun_out/r04d_reins_verbose.txt
timeout -k 10 900 python tests/tools/split_sweep.py c3 c3h c5 c2 --check --budgets -1 --slacks 3 --reinsert 0,2,4,8 > gpurun_out/r04d_reinsert_sweep.txt 2>&1; echo "sweep rc $?"; cat gpurun_out/r04d_reinsert_sweep.txt
