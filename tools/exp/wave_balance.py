"""Load balance of the quad rollout kernels (experiment, not a test): residency of every wavefront of one rollout at the
initial state and at the steady closed-loop state of a workload (mppi_set_wave_clock / mppi_get_wave_clock, 100 MHz ticks).
    python tools/exp/wave_balance.py [workload ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mppi-isaac_amd"))
import numpy as np

import bench
from mppiisaac.backend import capi


def report(tag, loop):
    lib, P = loop.lib, loop.P
    info = C.create_string_buffer(256)
    capi.check(lib, lib.mppi_kernel_info(P, info, 256))
    n = int(dict(kv.split("=") for kv in info.value.decode().split())["waves"])
    capi.check(lib, lib.mppi_set_wave_clock(P, 1))
    capi.check(lib, lib.mppi_rollout(P))
    clk = np.zeros((n, 2), np.uint64)
    capi.check(lib, lib.mppi_get_wave_clock(P, clk.ctypes.data_as(C.POINTER(C.c_uint64)), n))
    capi.check(lib, lib.mppi_set_wave_clock(P, 0))
    t0 = clk[:, 0].min()
    start, end = (clk[:, 0] - t0).astype(np.float64) * 1e-2, (clk[:, 1] - t0).astype(np.float64) * 1e-2   # us
    dur = end - start
    print(f"{tag}: {n} wavefronts, kernel span {end.max():.1f} us; start spread {start.max():.1f} us; wavefront residency "
          f"min {dur.min():.1f} / mean {dur.mean():.1f} / median {np.median(dur):.1f} / p95 {np.percentile(dur, 95):.1f} / max {dur.max():.1f} us; "
          f"mean/max = {dur.mean() / dur.max():.3f}")
    print(f"   started later than 10 % of the span: {(start > 0.1 * end.max()).mean():.3f} of the wavefronts")
    hist, edges = np.histogram(dur, bins=10)
    print("   residency histogram:", " ".join(f"{int(e)}:{h}" for h, e in zip(hist, edges[:-1])))
    slow = np.argsort(dur)[-5:][::-1]
    print("   slowest chunks:", [(int(c), round(float(dur[c]), 1)) for c in slow])
    return dur


if __name__ == "__main__":
    names = sys.argv[1:] or ["panda_reach", "boxer_push", "panda_pick"]
    env = dict(world_size=1, rank=0, local_rank=0, sharded=False, backend="nccl", action_sync=False)
    for name in names:
        loop = bench.Loop(name, int(os.environ.get("K_TOTAL", bench.WORKLOADS[name]["K"])), env)
        report(f"{name} initial state", loop)
        for _ in range(int(os.environ.get("STEPS", "200"))):
            loop.iterate()
        report(f"{name} closed loop", loop)
