for nc in 1.5 2.0 2.5 3.5; do echo "== node cost $nc"; SOL_NODE_COST=$nc python tests/tools/perf_quick.py c2 c3 --spp 64 --phases; done
