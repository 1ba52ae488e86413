#!/bin/bash
# round 4, call 24: the final build (rotated records, needle pad 4, background blocks) against the oracle: full frames and 2000 random scenes
set -o pipefail
mkdir -p gpurun_out
{ echo "Final build of round 4 (fp32 records rotated, needle pad 4 thin pads, background blocks summed instead of traced) against the fp32 oracle, FULL frames (tests/tools/gpu_full_oracle.py): C3 1080p x 32 spp, C2 1080p x 32 spp, C5 960x540 x 8 spp, the heterogeneous atrium 1080p x 16 spp, C1 400x400 x 50, the reference test scene 800x400 x 64 spp; then random scenes 11000.. (2000). bad = pixels over 1e-5 relative."
cd tests/tools
timeout -k 10 300 python gpu_full_oracle.py c3 32 1920 1080 default
timeout -k 10 300 python gpu_full_oracle.py c2 32 1920 1080 default
timeout -k 10 300 python gpu_full_oracle.py c5 8 960 540 default
timeout -k 10 300 python gpu_full_oracle.py c3h 16 1920 1080 default
timeout -k 10 300 python gpu_full_oracle.py c1 50 400 400 default
timeout -k 10 300 python gpu_full_oracle.py test 64 800 400 default
timeout -k 10 500 python random_parity_sweep.py 11000 2000
cd ../..; } > gpurun_out/call24_full_frame.txt 2>&1
tail -20 gpurun_out/call24_full_frame.txt
