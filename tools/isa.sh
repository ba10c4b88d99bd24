#!/bin/bash
# tools/isa.sh <file.hip> <kernel-substring>  -- dump gfx950 ISA of one kernel and summarise the
# wait/branch/memory skeleton (what the vmcnt discipline of DESIGN.md §3 is checked with).
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/maximilian_amd/csrc/$1"
PAT="$2"
OUT=/tmp/isa_$(basename "$1" .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -S \
    --cuda-device-only -I"$ROOT/maximilian_amd/csrc" "$SRC" -o "$OUT" 2>&1 | grep -v hip-link || true
grep -E "\.name:|\.private_segment_fixed_size|\.vgpr_count|\.sgpr_count" "$OUT" | grep -v amdhsa |
    paste - - - - | awk '{print $2, $4, $6, $8}' | sed 's/_ZN3mxg12_GLOBAL__N_1//' | sort -k1
if [ -n "$PAT" ]; then
    awk -v k="$PAT" 'index($0,k) && /^_Z.*:/{f=1} f{print} f&&/s_endpgm/{exit}' "$OUT" > /tmp/isa_kernel.s
    echo "--- $PAT: $(wc -l < /tmp/isa_kernel.s) lines -> /tmp/isa_kernel.s"
    grep -n -E "s_waitcnt vmcnt|s_cbranch|^\.LBB|global_load|global_store|scratch_" /tmp/isa_kernel.s |
        awk '{print $1,$2,$3}'
fi
