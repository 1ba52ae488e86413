python tests/tools/variants.py run --scenes "c2 c3" default pm6 pm12
for sw in 12 20 24; do echo "== switch $sw"; SOL_SWITCH=$sw python tests/tools/perf_quick.py c2 c3 --spp 64; done
