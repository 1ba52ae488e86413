set -o pipefail
mkdir -p gpurun_out
SOL_VERBOSE=1 SOL_REINSERT=6 timeout -k 10 600 python -m pytest tests/test_world_tree.py -q -m gpu -k "reference_test_scene or cornell or degenerate or chain" > gpurun_out/r04d_tree_tests.log 2>&1; rc=$?; echo "tree tests (6 reinsertion rounds) rc $rc"; grep -E "reinsertion|Error|passed|failed|DeviceError" gpurun_out/r04d_tree_tests.log | head -60
