for v in base default; do
  if [ $v = base ]; then export SOLSTRALE_BUILD_DIR=$PWD/_var/base; else unset SOLSTRALE_BUILD_DIR; fi
  echo "== $v"; SOL_VERBOSE=1 python tests/tools/perf_quick.py c2 c3 --spp 64 --phases 2>&1 | grep -v "work order"
done
