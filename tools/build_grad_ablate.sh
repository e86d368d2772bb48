#!/bin/bash
# builds copies of the library whose fused gradient pass leaves out a part (measurement only, wrong results):
# build_ab/librdis_hip_grad_ablate{1,2,4}.so -- 1: no factor arithmetic, 2: no segment sums, 4: neither (loads, rows, barriers)
set -e
cd /root/repo/rdis_amd/csrc
make -s >/dev/null
mkdir -p /root/repo/build_ab
for n in ${VARIANTS:-1 2 4}; do
  mkdir -p /root/repo/build_ab/obj_g$n
  cp /root/repo/rdis_amd/lib/obj/*.o /root/repo/build_ab/obj_g$n/
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DRDIS_GRAD_ABLATE=$n ${EXTRA:-} -c -o /root/repo/build_ab/obj_g$n/grad_fused.o grad_fused.hip &
done
wait
for n in ${VARIANTS:-1 2 4}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build_ab/librdis_hip_grad_ablate$n.so /root/repo/build_ab/obj_g$n/*.o
done
ls -la /root/repo/build_ab/librdis_hip_grad_ablate*.so
