#!/bin/bash
# diagnostic build: libka9q_hip_dbg.so = the product's objects with kh_fm2_kernels.hip recompiled under -DKH_FM2_TIMES
cd "$(dirname "$0")/../../ka9q-radio_amd/csrc" && make >/dev/null && rm -rf build_dbg && mkdir build_dbg && cp build/*.o build_dbg/ && \
/opt/rocm/bin/hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -fPIC -std=c++17 -DKH_FM2_TIMES -x hip -c kh_fm2_kernels.hip -o build_dbg/kh_fm2_kernels.hip.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libka9q_hip_dbg.so build_dbg/*.o -lpthread && rm -rf build_dbg
