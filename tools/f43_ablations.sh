#!/bin/bash
# Builds tools/f43_bench.hip once per ablation value of the LIBRARY header (-DF43_ABL=n; rerevst-code_amd/csrc/conv_f43.h)
# and runs the binaries: rates as shipped, then with parts of the kernel switched off, then the per-phase timeline.
#   bash tools/f43_ablations.sh build     (here: hipcc cross-compiles)      bash tools/f43_ablations.sh run   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
VARIANTS="0 4 32 33 34 40 41 43 16 20"     # 4 no stores | 32 no epilogue | +1 no LDS-DMA | +2 no barriers | +8 no input transform | 16 timeline | 20 timeline without stores
if [ "$1" = build ]; then
    for v in $VARIANTS; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DF43_ABL=$v tools/f43_bench.hip -o tools/bin/f43_bench_$v &
    done
    wait
else
    for v in $VARIANTS; do echo "== F43_ABL=$v"; tools/bin/f43_bench_$v; done
fi
