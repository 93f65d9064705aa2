#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -- an experimental build of the library, e.g.
#   tools/build_variant.sh lps16 -DTNCO_LPS=16      ->  build_variants/lib_lps16.so
# (compare with tools/bench_variants.sh, which runs bench.py once per library in build_variants/)
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build_variants"
make -C "$ROOT/tnco_amd/csrc" -j8 OBJDIR=build_$NAME EXTRA="$*" OUT=../../build_variants/lib_$NAME.so ../../build_variants/lib_$NAME.so
