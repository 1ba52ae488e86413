#!/bin/bash
# round 4, call 20: fp32 records rotated to start opposite the longest edge, needle pad 4 thin pads.
# Full GPU suite, the full frames against the oracle again (every triangle scene's fp32 frame changed), bench lines of the triangle workloads.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/call20_gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc $rc"; tail -4 gpurun_out/call20_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
{ echo "Build with rotated fp32 triangle records (start opposite the longest edge) and the needle pad of 4 thin pads, against the fp32 oracle, FULL frames (tests/tools/gpu_full_oracle.py): C3 1080p x 32 spp, the HETEROGENEOUS atrium 1080p x 16 spp (pre-split tree), C5 960x540 x 8 spp, the reference test scene 800x400 x 64 spp. bad = pixels over 1e-5 relative."
cd tests/tools
timeout -k 10 300 python gpu_full_oracle.py c3 32 1920 1080 default
timeout -k 10 300 python gpu_full_oracle.py c3h 16 1920 1080 default
timeout -k 10 300 python gpu_full_oracle.py c5 8 960 540 default
timeout -k 10 300 python gpu_full_oracle.py test 64 800 400 default
cd ../..; } > gpurun_out/call20_full_frame.txt 2>&1
tail -12 gpurun_out/call20_full_frame.txt
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" --no-build > gpurun_out/r04u_${name}_bench.json 2> gpurun_out/r04u_${name}_bench.err; echo "$name rc $?"; }
run c3 --cpu-seconds 5
run c3_heterogeneous --workload c3 --mesh-preset heterogeneous --no-cpu-baseline
run c3_heterogeneous_interior --workload c3 --mesh-preset heterogeneous --camera-preset interior --no-cpu-baseline
run c5 --workload c5 --no-cpu-baseline
for f in gpurun_out/r04u_*_bench.json; do python -c "
import json
d = json.load(open('$f'))
v = d.get('roofline_valu', {})
print('$f', d['value'], d['mrays_per_s'], d['ms_per_step'], 'alg', d['roofline']['frac'], 'lanes', v.get('lane_utilisation'), 'valu', v.get('frac'), 'n/r', d['node_visits_per_ray'], d['primitive_tests_per_ray'], d['world_tree']['builder'], d['world_tree']['strict_triangles'])"; done
