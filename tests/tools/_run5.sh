for b in 2 3 4; do echo "== max_bpc $b"; SOL_MAX_BPC=$b python tests/tools/perf_quick.py c2 c3 --spp 64; done
