#!/bin/bash
# Same-box A/B of two builds of the native libraries: boxes differ by +-8 %, so
# two variants are only comparable when they run on ONE box in ONE gpurun call.
#   tools/ab_build.sh A            # build the current tree into spmv_amd/lib_A
#   (edit the source)
#   tools/ab_build.sh B            # ... and into spmv_amd/lib_B
#   gpurun -- 'for v in A B A B; do SPMV_AMD_LIBDIR=$PWD/spmv_amd/lib_$v python tools/ksweep.py ...; done'
# Delete the lib_* directories afterwards (they travel with every gpurun push).
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
# optional 2nd argument: extra hipcc flags (e.g. -DSOME_PROBE=1); forces a rebuild
if [ -n "${2:-}" ]; then rm -f "$ROOT"/spmv_amd/lib/obj/spmv_*.o; fi
make -C "$ROOT/spmv_amd/csrc" -j8 HIPEXTRA="${2:-}" >/dev/null
mkdir -p "$ROOT/spmv_amd/lib_$1"
cp "$ROOT"/spmv_amd/lib/*.so "$ROOT/spmv_amd/lib_$1/"
echo "built variant $1 -> spmv_amd/lib_$1"
