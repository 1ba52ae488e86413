set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_world_tree.py -x -q -m gpu > gpurun_out/r04b_tree_tests.log 2>&1; rc=$?; echo "tree tests rc $rc"; tail -15 gpurun_out/r04b_tree_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python tests/tools/split_sweep.py c3 c3h c3hi c5 c2 --check --budgets 0,10,30,60 --slacks 2,3,4 > gpurun_out/r04b_split_sweep.txt 2>&1; echo "sweep rc $?"; cat gpurun_out/r04b_split_sweep.txt
