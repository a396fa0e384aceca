#!/bin/bash
# CPU test suite against the AddressSanitizer/UBSan build of the host mirror
# (plan construction, matrix splitting, PETSc reader, C facade).  Leak checking
# is off: the interpreter and torch hold allocations for the process lifetime.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
make -C "$ROOT/spmv_amd/csrc" asan -j8 >/dev/null
export SPMV_AMD_LIBDIR="$ROOT/gpurun_out/asan"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
cd "$ROOT"
exec python -m pytest tests -q -x -m "not gpu" "$@"
