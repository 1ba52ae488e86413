#!/bin/bash
# round 5: fine-tail length (SOL_FINE_TAIL, quarters of a whole item per resident lane) on the product kernel, every workload
cd "$(dirname "$0")"
for f in 0 8 16 32 64; do echo "== SOL_FINE_TAIL=$f (64 spp)"; SOL_FINE_TAIL=$f timeout -k 5 300 python perf_quick.py c1 c2 c3 c5 test --spp 64 || exit 1; done
for f in 0 8 32; do echo "== SOL_FINE_TAIL=$f (256 spp)"; SOL_FINE_TAIL=$f timeout -k 5 300 python perf_quick.py c2 test --spp 256 || exit 1; done
