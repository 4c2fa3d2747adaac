#!/bin/bash
# CPU sanitizer run: the wavefront-emulator build of the product's tree kernels + host engine
# and the CPU oracle, both compiled with -fsanitize=address,undefined, under the emulator / oracle tests.
#   bash tools/run_sanitized.sh                     # the whole emulator + oracle suites (several minutes)
#   bash tools/run_sanitized.sh tests/test_engine_emul.py::test_static     # a selection
# ASan's runtime has to be the first library of the process, so it is preloaded into python; leak checking is off
# (the interpreter itself leaks by design); UBSan findings abort (-fno-sanitize-recover).
set -e
cd "$(dirname "$0")/.."
make -s -C tests/emul SAN=1
make -s -C oracle oracle SAN=1
ASAN_LIB=$(g++ -print-file-name=libasan.so)
UBSAN_LIB=$(g++ -print-file-name=libubsan.so)
TARGETS=("$@")
[ ${#TARGETS[@]} -eq 0 ] && TARGETS=(tests/test_engine_emul.py tests/test_oracle_golden.py tests/test_dropin_emul.py)
FPC_SAN=1 LD_PRELOAD="$ASAN_LIB:$UBSAN_LIB" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python3 -m pytest -x -q -m "not gpu" -p no:cacheprovider "${TARGETS[@]}"
