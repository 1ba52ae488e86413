#!/bin/bash
# CPU-side sanitizer run (CPU build only - no GPU sanitizer runs on this pool): builds libsolstrale_host.so, the host code
# of libsolstrale_hip.so (what sol_world_tree_check, sol_scene_create's validation and the flattener reach without a GPU) and liboracle.so
# with -fsanitize=address,undefined into solstrale-rust_amd/_build_san/ and runs the CPU suites that exercise them - the host mirror,
# the OBJ + MTL loader with malformed inputs and with a 262 267-triangle file (tests/test_obj_scale.py), the tree builders, the oracle's KATs and goldens, the ABI checks, 3 000 mutated scene descriptors (tests/test_desc_mutations.py) - with the sanitizer
# runtime preloaded into Python. Any report fails the run (halt_on_error). Usage: bash tests/tools/sanitize.sh [pytest args]
set -eo pipefail
root="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$root"
python solstrale-rust_amd/build.py --sanitize > /dev/null 2> "${TMPDIR:-/tmp}/sanitize_build.log" || { tail -20 "${TMPDIR:-/tmp}/sanitize_build.log"; exit 1; }
san="$root/solstrale-rust_amd/_build_san"
rt="$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)"
export LD_PRELOAD="$rt"
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0:verify_asan_link_order=0:detect_odr_violation=0"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export SOLSTRALE_BUILD_DIR="$san" SOLSTRALE_ORACLE_LIB="$san/liboracle.so"
python -m pytest tests/test_host.py tests/test_obj_loader.py tests/test_world_tree.py tests/test_oracle_kat.py tests/test_oracle_golden.py tests/test_abi.py tests/test_fp32_contract.py tests/test_background_blocks.py tests/test_gpu_examples.py tests/test_obj_scale.py tests/test_desc_mutations.py tests/test_host_api_sequences.py \
  -q -m "not gpu" -p no:cacheprovider -k "not gfx950_code_object and not missing_communication_library" "$@"
