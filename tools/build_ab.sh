#!/bin/bash
# tools/build_ab.sh NAME FILE.hip "-DSWITCH=..."  -- an A/B build of libmaxigpu.so with ONE object recompiled under extra -D switches:
# maximilian_amd/ab_NAME.so (git-ignored, travels to the GPU box); select it with MXG_LIB (maximilian_amd/_lib.py).
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
cd $R/maximilian_amd/csrc
make -j8 > /dev/null
name=$1; src=$2; shift 2
obj=/tmp/ab_${name}_$(basename $src .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable "$@" -c $src -o $obj 2>&1 | grep -E "error" || true
objs=""
for o in *.o; do if [ "$o" = "$(basename $src .hip).o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab_$name.so $objs -ldl -Wl,-rpath,/opt/rocm/lib
ls -la ../ab_$name.so
