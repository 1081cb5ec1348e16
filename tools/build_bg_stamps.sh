#!/bin/bash
# Developer tool: experiment build of the library with the switches of the blocked contractions (-DBG_STAMPS: blocked.hip,
# GMMVI_BG_DEBUG bits: 1 no MFMAs, 2 no split + LDS store of A, 4 no global loads, 8 no LDS store of B, 64 / 128 fixed B / A
# addresses, 256 no staging priority) as gmmvi_amd/libgmmvi_hip_bgstamps.so; use it with GMMVI_HIP_LIB=... (tools/bg_debug_sweep.sh).
set -e
cd "$(dirname "$0")/../gmmvi_amd/csrc"
make -j8 >/dev/null
mkdir -p build_stamps
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -ffp-contract=fast \
    -DBG_STAMPS -c blocked.hip -o build_stamps/blocked.o
objs=$(ls build/*.o | grep -v blocked.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib $objs build_stamps/blocked.o \
    -o ../libgmmvi_hip_bgstamps.so
echo built gmmvi_amd/libgmmvi_hip_bgstamps.so
