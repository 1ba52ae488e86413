#!/bin/bash
# round 4, call 21: the round-3 builder's tree on the heterogeneous mesh under the final contract (rotated records, needle pad 4): the A/B base again
set -o pipefail
mkdir -p gpurun_out
SOL_SPLIT=0 SOL_REINSERT=0 SOL_PLOC_R=16 timeout -k 10 500 python bench.py --workload c3 --mesh-preset heterogeneous --no-cpu-baseline --no-pmc --no-build > gpurun_out/r04u_c3_heterogeneous_r03tree_bench.json 2> gpurun_out/r04u_c3_heterogeneous_r03tree_bench.err; echo "rc $?"
python -c "
import json
d = json.load(open('gpurun_out/r04u_c3_heterogeneous_r03tree_bench.json'))
print(d['value'], d['mrays_per_s'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['node_visits_per_ray'], d['primitive_tests_per_ray'], d['world_tree'])"
cd tests/tools
echo "== round-3 tree (SOL_SPLIT=0 SOL_REINSERT=0 SOL_PLOC_R=16)"; SOL_SPLIT=0 SOL_REINSERT=0 SOL_PLOC_R=16 timeout -k 10 300 python perf_quick.py c3h --spp 64 --phases
echo "== no pre-splitting (SOL_SPLIT=0)"; SOL_SPLIT=0 timeout -k 10 300 python perf_quick.py c3h --spp 64 --phases
echo "== default"; timeout -k 10 300 python perf_quick.py c3h c3 c5 --spp 64 --phases
