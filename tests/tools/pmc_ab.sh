#!/bin/bash
# A/B of two environment settings under one counter group: bash pmc_ab.sh "<counters>" VAR=a VAR=b   (run from the repo root)
export TMPDIR=/tmp
out=$PWD/gpurun_out
grp=$1; shift
for setting in "$@"; do
  name=$(echo "$setting" | tr '= ' '__')
  (cd /tmp && env "$setting" true; export $setting; timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/ab_pmc/$name/grp -- python3 $OLDPWD/bench.py --spp 64 --steps 1 --warmup 0 --no-cpu-baseline > $out/ab_pmc_$name.log 2>&1) || echo "pass failed $setting"
done
