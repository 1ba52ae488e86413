set -o pipefail
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" --no-build > gpurun_out/r04t_${name}_bench.json 2> gpurun_out/r04t_${name}_bench.err; echo "$name rc $?"; }
run c3 --cpu-seconds 5
run c3_heterogeneous --workload c3 --mesh-preset heterogeneous --no-cpu-baseline
run c3_heterogeneous_interior --workload c3 --mesh-preset heterogeneous --camera-preset interior --no-cpu-baseline
SOL_SPLIT=0 SOL_REINSERT=0 SOL_PLOC_R=16 run c3_heterogeneous_r03tree --workload c3 --mesh-preset heterogeneous --no-cpu-baseline --no-pmc
for f in gpurun_out/r04t_*_bench.json; do python -c "
import json
d = json.load(open('$f'))
v = d.get('roofline_valu', {})
print('$f', d['value'], d['mrays_per_s'], d['ms_per_step'], 'alg', d['roofline']['frac'], 'fabric', d['roofline'].get('traffic_frac_range'), 'lanes', v.get('lane_utilisation'), 'issue', v.get('issue_busy_at_4_cycles_per_instr'), 'valu', v.get('frac'), 'n/r', d['node_visits_per_ray'], d['primitive_tests_per_ray'], d['world_tree']['builder'], d['world_tree']['strict_triangles'])"; done
