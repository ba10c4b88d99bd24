#!/bin/bash
# tools/build_ab.sh NAME FILE.hip "-DSWITCH=..."  -- an A/B build of libmaxigpu.so with ONE object recompiled under extra -D switches:
# build/ab/ab_NAME.so (git-ignored, OUTSIDE the package directory; travels to the GPU box while it exists -- delete build/ab when the
# comparison is done); select it with MXG_LIB=build/ab/ab_NAME.so (maximilian_amd/_lib.py).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
cd $R/maximilian_amd/csrc
make -j8 > /dev/null
name=$1; src=$2; shift 2
obj=/tmp/ab_${name}_$(basename $src .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable $(make -s -f - flags_$(basename $src .hip) <<'MK'
include Makefile
flags_%: ; @echo $(FLAGS_$*)
MK
) "$@" -c $src -o $obj 2>&1 | grep -E "error" || true
objs=""
for o in *.o; do if [ "$o" = "$(basename $src .hip).o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi; done
mkdir -p $R/build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/ab/ab_$name.so $objs -ldl -Wl,-rpath,/opt/rocm/lib
ls -la $R/build/ab/ab_$name.so
