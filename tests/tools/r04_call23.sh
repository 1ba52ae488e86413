#!/bin/bash
# round 4, call 23: background blocks (lens cameras too), bench accounting: full GPU suite, headline + C5 bench lines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/call23_gpu_tests.log 2>&1; rc=$?; echo "gpu tests rc $rc"; tail -4 gpurun_out/call23_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" --no-build > gpurun_out/r04v_${name}_bench.json 2> gpurun_out/r04v_${name}_bench.err; echo "$name rc $?"; }
run c3 --cpu-seconds 5
run c5 --workload c5 --no-cpu-baseline
run c3_heterogeneous --workload c3 --mesh-preset heterogeneous --no-cpu-baseline
for f in gpurun_out/r04v_*_bench.json; do python -c "
import json
d = json.load(open('$f'))
v = d.get('roofline_valu', {})
print('$f', d['value'], d['mrays_per_s'], d['ms_per_step'], 'alg', d['roofline']['frac'], 'lanes', v.get('lane_utilisation'), 'valu', v.get('frac'), 'n/r', d['node_visits_per_ray'], d['primitive_tests_per_ray'], 'r/s', d['rays_per_sample'], d['traced_rays_per_sample'], d['background_blocks']['blocks'], d['background_blocks']['sample_fraction'], d['setup_s'])"; done
