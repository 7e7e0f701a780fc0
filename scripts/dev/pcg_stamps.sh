#!/bin/bash
# Diagnostic build of libaar with s_memtime stamps in k_pcgf (-DAAR_PCG_STAMPS) -> automatic-ar_amd/libaar_st.so; run on the GPU box:
#   bash scripts/dev/pcg_stamps.sh build            (here: cross-compiles)
#   AAR_LIB=$PWD/automatic-ar_amd/libaar_st.so python scripts/dev/pcg_stamps.py     (GPU box)
set -e
cd "$(dirname "$0")/../../automatic-ar_amd"
mkdir -p build_st/csrc build_st/host
for f in host/dataset.cpp host/synth.cpp host/solution_io.cpp host/multicam_mapper.cpp host/cam_config.cpp host/initializer.cpp host/host_levmarq.cpp; do cp build/${f%.cpp}.o build_st/${f%.cpp}.o; done
for f in hostcopy eval_kernels solve_kernels spcg_kernels ba_capi undistort init_kernels; do cp build/csrc/$f.o build_st/csrc/$f.o; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result --offload-arch=gfx950 -munsafe-fp-atomics -DAAR_PCG_STAMPS -c csrc/pcg_kernels.hip -o build_st/csrc/pcg_kernels.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libaar_st.so $(find build_st -name '*.o') -ldl -Wl,-rpath,/opt/rocm/lib
ls -la libaar_st.so
