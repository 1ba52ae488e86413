#!/bin/bash
# round 4, call 22: background blocks (blocks proved to see only the background are summed, not traced): parity, then A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "background_blocks or c5 or c3_sponza or tile_partition or ragged or sample_ranges or chunk_size or full_size_c3 or aux" > gpurun_out/call22_tests.log 2>&1
echo "tests rc $?"; tail -5 gpurun_out/call22_tests.log
cd tests/tools
echo "== default"; timeout -k 10 300 python perf_quick.py c2 c3 c3h c5 --spp 64
echo "== SOL_BACKGROUND_BLOCKS=0"; SOL_BACKGROUND_BLOCKS=0 timeout -k 10 300 python perf_quick.py c2 c3 c3h c5 --spp 64
