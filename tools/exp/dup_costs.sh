#!/bin/bash
# Cost of the sections of the contact pair loop by duplication (csrc/mppi_scene.hpp, MPPI_DUP): build the variants with
#   for k in 1 2 3 4 5; do MPPI_BUILD_VARIANT=dup$k python __graft_entry__.py; done
# then run this on the GPU box: kernel time of every variant at the recorded closed-loop states; difference to the product build
# = time of that section (1 preamble: pair record + shape poses + noise draws, 2 broad phase, 3 feature points / contact point,
# 4 cross-lane sum, 5 accumulation into the LDS rows).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "" dup1 dup2 dup3 dup4 dup5; do
  L=""; [ -n "$v" ] && L=$PWD/mppi-isaac_amd/csrc/libmppi_hip_$v.so
  echo "== variant: ${v:-product}"
  MPPI_HIP_LIB=$L CLOSED_LOOP_ONLY=1 python tools/exp/scene_breakdown.py 2>&1 | grep "full (seeded"
done
