# The round's profile runs on the GPU box (through gpurun, from the repo root):  bash tests/tools/round_profiles.sh <tag> [part]
#   part a: the headline - bench.py default (C3) + rocprofv3 --kernel-trace --stats of the same command + the full set of --pmc passes
#   part b: every other workload with its own --pmc passes (bench.py runs them: roofline.traffic, roofline_valu), the stress variants,
#           the reference's profiling workload
# Everything lands in gpurun_out/<tag>_*; the summaries to keep are copied into profiles/ afterwards.
set -o pipefail
tag=${1:-rXX}
part=${2:-ab}
if [[ $part == *a* ]]; then
  bash tests/tools/profile_round.sh $tag stats
  bash tests/tools/profile_round.sh $tag pmc > gpurun_out/${tag}_pmc.log 2>&1; tail -2 gpurun_out/${tag}_pmc.log
fi
if [[ $part == *b* ]]; then
  run() {  # name, bench arguments
    name=$1; shift
    timeout -k 10 500 python bench.py "$@" --cpu-seconds 5 --no-build > gpurun_out/${tag}_${name}_bench.json 2> gpurun_out/${tag}_${name}_bench.err; echo "$name rc $?"
  }
  run c1 --workload c1
  run c2 --workload c2
  run c4 --workload c4 --steps 2
  run c5 --workload c5 --steps 2
  run c5_pmc512 --workload c5 --steps 1 --warmup 0 --pmc-spp 512 --no-cpu-baseline   # (a launch 8 times longer under the counters: its tail weighs an eighth)
  run c5_hdri --workload c5 --hdri --steps 2 --no-cpu-baseline
  run c3_interior --workload c3 --camera-preset interior --no-cpu-baseline
  run c5_closeup --workload c5 --camera-preset closeup --steps 2 --no-cpu-baseline
  run profiling --workload profiling
  run c3_heterogeneous --workload c3 --mesh-preset heterogeneous --no-cpu-baseline          # the STRESS mesh, default tree (pre-split + reinsertion)
  run c3_heterogeneous_interior --workload c3 --mesh-preset heterogeneous --camera-preset interior --no-cpu-baseline
  SOL_SPLIT=0 SOL_REINSERT=0 run c3_heterogeneous_r03tree --workload c3 --mesh-preset heterogeneous --no-cpu-baseline --no-pmc   # the same with round 3's builder (A/B base)
  SOL_SPLIT=0 SOL_REINSERT=0 run c3_r03tree --workload c3 --no-cpu-baseline --no-pmc
  for f in gpurun_out/${tag}_*_bench.json; do python -c "
import json
d = json.load(open('$f'))
v = d.get('roofline_valu', {})
print('$f', d['value'], d['mrays_per_s'], d['ms_per_step'], 'alg', d['roofline']['frac'], 'fabric', d['roofline'].get('traffic_frac'), 'lanes', v.get('lane_utilisation'), 'issue', v.get('issue_busy_at_4_cycles_per_instr'), 'rays/sample', d['rays_per_sample'], 'primary', d.get('primary_hit_fraction'))"; done
fi
