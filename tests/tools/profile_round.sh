#!/bin/bash
# Profiling recipe of a round (run on the GPU box through gpurun, from the repo root):
#   bash tests/tools/profile_round.sh <tag>          -> gpurun_out/<tag>_{bench.json, bench_under_rocprof.json, kernel_stats.csv}
#   bash tests/tools/profile_round.sh <tag> pmc      -> gpurun_out/<tag>_pmc/<group>/... (one rocprofv3 --pmc pass per counter group)
# The summaries are then copied into profiles/ (tracked).
set -o pipefail
tag=${1:-rXX}
mode=${2:-stats}
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
if [ "$mode" = "stats" ]; then
  python3 bench.py --steps 3 --warmup 1 > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 1
  tail -c 600 $out/${tag}_bench.json
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_prof -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-build --no-all-traced \
      > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_rocprof.err) || exit 1
  find $out/${tag}_prof -name '*kernel_stats.csv' -exec cp {} $out/${tag}_kernel_stats.csv \;
  head -5 $out/${tag}_kernel_stats.csv
else
  # at most two TA/TD/TCP counters per pass (more: "Request exceeds the capabilities of the hardware")
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr" "FETCH_SIZE" "WRITE_SIZE"; do   # FETCH_SIZE takes 3 of the 4 TCC counters, WRITE_SIZE 2; TD_TD_BUSY says nothing (profiles/r04_counter_questions.txt)
    if [ -n "$3" ] && [[ "$grp" != *"$3"* ]]; then continue; fi
    name=$(echo $grp | tr ' ' '+')
    echo "== $grp"
    (cd /tmp && timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/${tag}_pmc/$name -- python3 $OLDPWD/bench.py --spp 64 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-build --no-all-traced \
        > $out/${tag}_pmc_$name.log 2>&1) || { echo "pass failed: $grp"; tail -3 $out/${tag}_pmc_$name.log; }
  done
fi
