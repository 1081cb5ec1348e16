#!/bin/bash
# Developer tool: experiment build of the library with in-kernel time stamps (-DGMMVI_ME_STAMPS: density.hip) as
# gmmvi_amd/libgmmvi_hip_stamps.so; use it with GMMVI_HIP_LIB=gmmvi_amd/libgmmvi_hip_stamps.so (tools/pk_stamps.py).
set -e
cd "$(dirname "$0")/../gmmvi_amd/csrc"
make -j8 >/dev/null
mkdir -p build_stamps
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function -ffp-contract=fast \
    -DGMMVI_ME_STAMPS $EXTRA_DEFS -c density.hip -o build_stamps/density.o
objs=$(ls build/*.o | grep -v density.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib $objs build_stamps/density.o \
    -o ../libgmmvi_hip_stamps.so
echo built gmmvi_amd/libgmmvi_hip_stamps.so
