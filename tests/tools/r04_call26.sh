#!/bin/bash
# round 4, call 26: more reinsertion rounds (1080p x 64 spp)
set -o pipefail
cd tests/tools
for r in 8 16 24 32 48; do echo "== SOL_REINSERT=$r"; SOL_VERBOSE=1 SOL_REINSERT=$r timeout -k 10 300 python perf_quick.py c2 c3 c3h c5 --spp 64 --phases 2>&1 | grep -v "device tree\|world tree probe\|background\|work order"; done
