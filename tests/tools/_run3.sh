python tests/tools/variants.py run --scenes "c2 c3" default pm1 pm4 pm12
for sw in 8 24 32; do echo "== switch $sw"; SOL_SWITCH=$sw python tests/tools/perf_quick.py c2 c3 --spp 64; done
