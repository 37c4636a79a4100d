#!/usr/bin/env python3
"""Life of every workgroup of one staged list build (ab/libsph_bnl_timeline.so from tools/removal_build.py bnl_timeline): thread 0's clock at the
phase boundaries of k_build_nl -- the cell set, the plan (local bases, cell runs, tile row), the three dx planes of the walk, the epilogue.

    python tools/removal_build.py bnl_timeline && python tools/bnl_timeline.py [scene] [advance_steps]"""
import ctypes
import json
import os
os.environ.setdefault("SPH_DEV", "1")
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SPH_LIB"] = os.path.join(ROOT, "ab", "libsph_bnl_timeline.so")
from cfd_taichi_amd import _native as nat, scenes  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "dfsph_1m"
advance = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
sim.step_dfsph(advance)
for _ in range(2):
    us = sim.tune_time(3, 0, 1)               # sort + list build of the current positions
nwg = (sim.n_fluid + 255) // 256
buf = np.zeros((nwg, 8), dtype=np.uint64)
lib = nat.load()
lib.sph_debug_bnl.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.sph_debug_bnl(buf.ctypes.data, nwg) == 0
t = buf[:, :7].astype(np.int64)
base = t[:, 0].min()
t = (t - base) / 100.0                          # wall_clock64: 100 MHz -> us
names = ["cell set (hash inserts, barrier)", "plan: bases, cell runs, tile row (two barriers)", "walk dx = -1 (table + 9 cells)", "walk dx = 0", "walk dx = +1", "flush, counts, maxima"]
d = np.diff(t, axis=1)
life = t[:, 6] - t[:, 0]
nruns = (buf[:, 7] >> np.uint64(32)).astype(np.int64)
out = {"scene": scene, "step": advance + 1, "workgroups": int(nwg), "sort_plus_build_us_events": us, "span_us": float(t[:, 6].max()),
       "life_us": {"mean": float(life.mean()), "p10": float(np.percentile(life, 10)), "p50": float(np.percentile(life, 50)), "p90": float(np.percentile(life, 90)), "max": float(life.max())},
       "phases_us_mean": {n: float(d[:, k].mean()) for k, n in enumerate(names)},
       "phases_us_p90": {n: float(np.percentile(d[:, k], 90)) for k, n in enumerate(names)},
       "phase_share_of_life": {n: float(d[:, k].sum() / life.sum()) for k, n in enumerate(names)},
       "wave0_runs_of_equal_cell": {"mean": float(nruns.mean()), "p90": float(np.percentile(nruns, 90)), "over_table": float((nruns > 13).mean())},
       "sum_life_over_span_x_resident": float(life.sum() / (t[:, 6].max() * 256 * 7))}
grid = np.arange(0, t[:, 6].max(), 5.0)
out["in_flight_every_5us"] = [int(((t[:, 0] <= x) & (t[:, 6] > x)).sum()) for x in grid]
print(json.dumps(out))
