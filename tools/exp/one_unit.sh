#!/bin/bash
# compile ONE generated unit of the library into /tmp and print its kernels' resources: tools/exp/one_unit.sh topo_5_scene [filter] [extra flags]
U=$1; F=${2:-k_rollout_scene_quad}; shift; shift
cd /root/repo/mppi-isaac_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c -o /tmp/$U.o $U.hip && python /root/repo/tools/kernel_stats.py /tmp/$U.o $F
