#!/bin/bash
# round 4, call 18: vertex-rotated fp32 records - parity subset, then the needle pad factor (1, 4, 40 thin pads) on the heterogeneous mesh
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fp32_records or needle or heterogeneous or uv_wrapping or c1_cornell or c3_sponza or obj or golden or random_scenes or normal_mapping or reference_test_scene" > gpurun_out/call18_tests.log 2>&1
echo "tests rc $?" | tee -a gpurun_out/call18_tests.log
tail -3 gpurun_out/call18_tests.log
timeout -k 10 600 python tests/tools/variants.py run --spp 64 --scenes "c3h c3" default pad4 pad1 > gpurun_out/call18_pad.log 2>&1
cat gpurun_out/call18_pad.log
