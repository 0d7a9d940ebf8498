#!/bin/bash
# Ablation builds of conv_units_fprop_kernel (TBG_EXP in csrc/conv_units.hip) -> tools/scratch/libexp<N>.so; then, on the GPU box,
# tools/exp_units_fprop.py times each on the largest layer.  Informs DESIGN 4.1c (what keeps the kernel's MFMA-busy at 0.64).
cd "$(dirname "$0")/.." || exit 1
OBJ=textboxgan_amd/csrc/.obj
for e in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DTBG_EXP=$e -c textboxgan_amd/csrc/conv_units.hip -o tools/scratch/cu_exp$e.o 2>/dev/null &
done
wait
for e in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/scratch/libexp$e.so tools/scratch/cu_exp$e.o $(ls $OBJ/*.o | grep -v conv_units)
done
ls -la tools/scratch/libexp*.so
