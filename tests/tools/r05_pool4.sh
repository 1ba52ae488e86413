#!/bin/bash
cd "$(dirname "$0")"
echo "== v1"; SOL_KERNEL=v1 timeout -k 5 300 python phase_counts.py c3 c2 test || exit 1
for sw in 8 24 40; do echo "== v4 SOL_POOL_SWAP=$sw"; SOL_KERNEL=v4 SOL_POOL_COUNT=1 SOL_POOL_SWAP=$sw timeout -k 5 300 python phase_counts.py c3 c2 test || exit 1; done
