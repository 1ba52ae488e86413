#!/bin/bash
cd "$(dirname "$0")"
for k in v1 v4; do echo "== SOL_KERNEL=$k (64 spp)"; SOL_KERNEL=$k timeout -k 5 300 python perf_quick.py c1 c2 c3 c3h test --spp 64 || exit 1; done
for k in v1 v4; do echo "== SOL_KERNEL=$k (test scene 256 spp, c3 128)"; SOL_KERNEL=$k timeout -k 5 300 python perf_quick.py test --spp 256 || exit 1; SOL_KERNEL=$k timeout -k 5 300 python perf_quick.py c3 --spp 128 || exit 1; done
for sw in 8 24; do echo "== v4 SOL_POOL_SWAP=$sw"; SOL_KERNEL=v4 SOL_POOL_SWAP=$sw timeout -k 5 300 python perf_quick.py c2 c3 test --spp 64 || exit 1; done
