#!/bin/bash
# builds a -DRDIS_COOP_TIMING copy of the library into build_ab/librdis_hip_timing.so
set -e
cd /root/repo/rdis_amd/csrc
mkdir -p /root/repo/build_ab/obj_t
for f in ${FILES:-rdis_hip ptm_kernels grad_fused refround_kernels components lm_solver}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DRDIS_COOP_TIMING -c -o /root/repo/build_ab/obj_t/$f.o $f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build_ab/librdis_hip_timing.so /root/repo/build_ab/obj_t/*.o
ls -la /root/repo/build_ab/librdis_hip_timing.so
