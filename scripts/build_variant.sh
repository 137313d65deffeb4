#!/bin/bash
# Builds a kernel-variant copy of the engine for A/B runs on the GPU box (selected with LANCE_HIP_LIB=<path>):
#   scripts/build_variant.sh NAME [file.hip=<git-ref>|file.hip=<path>] ... [-- extra hipcc flags]
# Every object not named is taken from the regular build (build/obj); named sources are compiled into build/variants/NAME/
# (sources from a git ref are checked out into that directory first).  Output: build/variants/liblance_hip_NAME.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
OUT=$ROOT/build/variants/$NAME
mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
EXTRA=""
declare -A REPL
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; EXTRA="$*"; break; fi
  f=${1%%=*}; src=${1#*=}
  if [ "$f" == "$src" ]; then src=$ROOT/lance_amd/csrc/$f          # recompile the tree's own source (with the extra flags)
  elif [ ! -f "$src" ]; then git -C "$ROOT" show "$src:lance_amd/csrc/$f" > "$OUT/$f"; src=$OUT/$f; fi
  REPL[$f]=$src
  shift
done
OBJS=""
for o in "$ROOT"/build/obj/*.o; do
  b=$(basename "$o" .o)
  if [ -n "${REPL[$b]}" ]; then
    hipcc $FLAGS $EXTRA -I"$ROOT/lance_amd/csrc" -x hip -c "${REPL[$b]}" -o "$OUT/$b.o"
    OBJS="$OBJS $OUT/$b.o"
  else
    OBJS="$OBJS $o"
  fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build/variants/liblance_hip_$NAME.so" $OBJS -ldl -pthread
echo "built build/variants/liblance_hip_$NAME.so"
