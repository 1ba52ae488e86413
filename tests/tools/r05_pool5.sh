#!/bin/bash
cd "$(dirname "$0")"
echo "== v1"; SOL_KERNEL=v1 timeout -k 5 300 python perf_quick.py c2 c3 test --spp 64 || exit 1
for sw in 16 24 32 48 64; do echo "== v4 SOL_POOL_SWAP=$sw"; SOL_KERNEL=v4 SOL_POOL_SWAP=$sw timeout -k 5 300 python perf_quick.py c2 c3 test --spp 64 || exit 1; done
cd ../..
SOL_KERNEL=v4 SOL_POOL_SWAP=24 SOL_BENCH_KERNEL=sol_render_pool4 python bench.py --spp 64 --steps 3 --no-cpu-baseline --no-all-traced > gpurun_out/r05_pmc_v4_k24.json 2> gpurun_out/r05_pmc_v4_k24.err
