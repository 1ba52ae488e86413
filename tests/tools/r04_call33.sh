#!/bin/bash
# round 4, call 33: triangle lights intersected through rotated records, sampled in the reference's frame (DevScene::light_tri)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/call33_gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc $rc"; tail -4 gpurun_out/call33_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
cd tests/tools
timeout -k 10 300 python perf_quick.py c1 c2 c3 c3h c5 test --spp 64
timeout -k 10 300 python perf_quick.py c3 c5 --spp 64
timeout -k 10 600 python needle_parity_sweep.py 6000 2000
timeout -k 10 500 python random_parity_sweep.py 30000 2000
