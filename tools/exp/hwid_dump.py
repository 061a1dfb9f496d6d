"""Experiment (MPPI_BUILD_VARIANT=hwid): which SIMD / wave slot the two wavefronts of the helper-wavefront kernel's workgroups
land on.  python tools/exp/hwid_dump.py   (MPPI_HIP_LIB=.../libmppi_hip_hwid.so)"""
import ctypes as C
import os
import sys
from collections import Counter

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import numpy as np

import bench
from mppiisaac.backend import capi

env = dict(world_size=1, rank=0, local_rank=0, sharded=False, backend="nccl", action_sync=False)
loop = bench.Loop("boxer_push", 8192, env)
lib, P = loop.lib, loop.P
n = 1024
capi.check(lib, lib.mppi_set_wave_clock(P, 1))
capi.check(lib, lib.mppi_rollout(P))
clk = np.zeros((n, 2), np.uint64)
capi.check(lib, lib.mppi_get_wave_clock(P, clk.ctypes.data_as(C.POINTER(C.c_uint64)), n))
v = clk[:, 0]
h0, h1 = (v & 0xFFFFFFFF).astype(np.uint32), (v >> 32).astype(np.uint32)
f = lambda h: ((h >> 4) & 3, h & 15, (h >> 8) & 15, (h >> 13) & 7)   # SIMD, slot, CU, SE
print("chunk: (simd, slot, cu, se) of wavefront 0 | wavefront 1")
for c in range(0, 40):
    print(c, [int(x) for x in f(h0[c])], [int(x) for x in f(h1[c])])
print("simd pairs:", Counter(zip(f(h0)[0].tolist(), f(h1)[0].tolist())).most_common(8))
print("slot pairs:", Counter(zip(f(h0)[1].tolist(), f(h1)[1].tolist())).most_common(8))
