#!/bin/bash
# round 4, call 25: collapse cost and reinsertion rounds re-swept on the final build (1080p x 64 spp)
set -o pipefail
cd tests/tools
for nc in 1.5 2.0 2.5 3.0 4.0; do echo "== SOL_NODE_COST=$nc"; SOL_NODE_COST=$nc timeout -k 10 300 python perf_quick.py c2 c3 c5 --spp 64 --phases; done
for r in 4 16; do echo "== SOL_REINSERT=$r"; SOL_REINSERT=$r timeout -k 10 300 python perf_quick.py c2 c3 c5 --spp 64 --phases; done
