#!/bin/bash
# round 5: first runs of the pool kernel (SOL_KERNEL=v4) against the product kernel: frame CRCs must agree
cd "$(dirname "$0")"
echo "== small frames first (a hang must cost seconds)"
SOL_KERNEL=v1 timeout -k 5 120 python perf_quick.py c1 test --spp 16 || exit 1
SOL_KERNEL=v4 timeout -k 5 120 python perf_quick.py c1 test --spp 16 || exit 1
for k in v1 v4 v1 v4; do echo "== SOL_KERNEL=$k"; SOL_KERNEL=$k timeout -k 5 300 python perf_quick.py c1 c2 c3 c3h c5 test --spp 64 || exit 1; done
