#!/bin/bash
# Builds orca_amd/liborca_hip_diag.so: the product library with tgat.hip compiled -DORCA_WAVE_STAMPS (every wave of the
# attention kernel stamps the shader clock around the GEMM stages when orca_debug_tgat_phase_clock is armed).
# Use: ORCA_HIP_LIB=$PWD/orca_amd/liborca_hip_diag.so python tools/kbench.py --attention-only
set -e
cd "$(dirname "$0")/.."
python -m orca_amd.build > /dev/null
cd orca_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++20 -fPIC -fvisibility=hidden -I../../include -DORCA_WAVE_STAMPS \
  -c tgat.hip -o build/tgat_diag.o
objs=$(ls build/*.o | grep -v -e tgat_diag.o -e '/tgat.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liborca_hip_diag.so $objs build/tgat_diag.o
echo orca_amd/liborca_hip_diag.so
