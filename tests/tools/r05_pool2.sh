#!/bin/bash
# round 5: pool kernel (SOL_KERNEL=v4) parameter sweep: swap batch (SOL_POOL_SWAP) x switch threshold (SOL_SWITCH)
cd "$(dirname "$0")"
echo "== v1"; SOL_KERNEL=v1 timeout -k 5 300 python perf_quick.py c1 c2 c3 test --spp 64 || exit 1
for sw in 16 32 48; do for k in 8 16 32; do
  echo "== v4 SOL_POOL_SWAP=$k SOL_SWITCH=$sw"; SOL_KERNEL=v4 SOL_POOL_SWAP=$k SOL_SWITCH=$sw timeout -k 5 300 python perf_quick.py c1 c2 c3 test --spp 64 || exit 1
done; done
