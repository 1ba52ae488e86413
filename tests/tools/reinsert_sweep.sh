#!/bin/bash
# reinsertion rounds of the GPU tree build (SOL_REINSERT) against render time and creation time
cd "$(dirname "$0")"
for r in 8 16 32; do echo "== SOL_REINSERT=$r"; SOL_REINSERT=$r timeout -k 5 600 python perf_quick.py c2 c3 c3h c5 --spp 64 || exit 1; SOL_REINSERT=$r python create_only.py c5 | tail -1; SOL_REINSERT=$r python create_only.py c3 | tail -1; done
