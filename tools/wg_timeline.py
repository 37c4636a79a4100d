#!/usr/bin/env python3
"""Life of every workgroup of one divergence-residual launch (ab/libsph_wg_timeline.so from tools/removal_build.py wg_timeline):
how many workgroups are in flight over the launch, how long they live, per XCD when the last one ends.

    python tools/removal_build.py wg_timeline && python tools/wg_timeline.py [scene] [advance_steps]
    python tools/removal_build.py wg_timeline_rx && SPH_ARITH=relaxed python tools/wg_timeline.py        (the relaxed sweep)"""
import ctypes
import json
import os
os.environ.setdefault("SPH_DEV", "1")     # tools run with development overrides enabled (sph_overrides reports them)
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SPH_LIB"] = os.path.join(ROOT, "ab", "libsph_wg_timeline_rx.so" if os.environ.get("SPH_ARITH", "").startswith("r") else "libsph_wg_timeline.so")
from cfd_taichi_amd import _native as nat, scenes  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "dfsph_1m"
advance = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
sim.step_dfsph(advance)
sim.build_neighbors()
for _ in range(3):
    us = sim.tune_time(0, 0, 1)
nwg = (sim.n_fluid + 255) // 256
buf = np.zeros((nwg, 4), dtype=np.uint64)
lib = nat.load()
lib.sph_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.sph_debug_timeline(buf.ctypes.data, nwg) == 0
t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 1].astype(np.int64); tile = buf[:, 3].astype(np.int64)
w2 = buf[:, 2]
xcc = (w2 & np.uint64(0xff)).astype(np.int64)
ph_staged = ((w2 >> np.uint64(8)) & np.uint64(0xffff)).astype(np.float64) / 100.0      # us after the workgroup's begin (thread 0; barriers added in front of the stamps)
ph_pairs = ((w2 >> np.uint64(24)) & np.uint64(0xffff)).astype(np.float64) / 100.0
ph_walls = ((w2 >> np.uint64(40)) & np.uint64(0xffff)).astype(np.float64) / 100.0
base = t0.min()
b, e = (t0 - base) / 100.0, (t1 - base) / 100.0          # wall_clock64: 100 MHz -> us
life = e - b
out = {"scene": scene, "step": advance + 1, "workgroups": int(nwg), "launch_us_events": us, "span_us": float(e.max()),
       "life_us": {"mean": float(life.mean()), "p10": float(np.percentile(life, 10)), "p50": float(np.percentile(life, 50)),
                   "p90": float(np.percentile(life, 90)), "p99": float(np.percentile(life, 99)), "max": float(life.max())},
       "sum_life_over_span_x_slots": float(life.sum() / (e.max() * 1024)),
       "phases_us_mean": {"staging (begin -> operands in LDS)": float(ph_staged.mean()), "fluid pair loop (all four waves done)": float((ph_pairs - ph_staged).mean()),
                          "wall terms": float((ph_walls - ph_pairs).mean()), "epilogue + block partial": float((life - ph_walls).mean())},
       "phases_us_p90": {"staging": float(np.percentile(ph_staged, 90)), "pairs": float(np.percentile(ph_pairs - ph_staged, 90)), "walls": float(np.percentile(ph_walls - ph_pairs, 90)),
                         "epilogue": float(np.percentile(life - ph_walls, 90))},
       "per_xcd": {}}
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    out["per_xcd"][str(x)] = {"workgroups": int(m.sum()), "first_begin_us": float(b[m].min()), "last_end_us": float(e[m].max()), "sum_life_us": float(life[m].sum())}
if hasattr(lib, "sph_debug_sub"):              # the relaxed build also stamps the sub-phases of the staging (thread 0's clock)
    sub = np.zeros((nwg, 4), dtype=np.uint64)
    lib.sph_debug_sub.argtypes = [ctypes.c_void_p, ctypes.c_int]
    if lib.sph_debug_sub(sub.ctypes.data, nwg) == 0:
        ok = (sub[:, 0] >= buf[:, 0]) & (sub[:, 2] >= sub[:, 0]) & (sub[:, 2] <= buf[:, 1])          # staged workgroups of THIS launch
        s1 = ((sub[:, 0].astype(np.int64) - t0) / 100.0)[ok]; s2 = ((sub[:, 1].astype(np.int64) - t0) / 100.0)[ok]; s3 = ((sub[:, 2].astype(np.int64) - t0) / 100.0)[ok]
        out["staging_us_mean"] = {"workgroups": int(ok.sum()), "count + cell runs read, list expanded in LDS (barrier)": float(s1.mean()), "own indices taken (barrier)": float((s2 - s1).mean()),
                                  "gathers returned, operands stored to LDS (thread 0)": float((s3 - s2).mean()), "last barrier (all threads' stores)": float((ph_staged[ok] - s3).mean())}
if hasattr(lib, "sph_debug_sub2"):
    sub2 = np.zeros((nwg, 4), dtype=np.uint64)
    lib.sph_debug_sub2.argtypes = [ctypes.c_void_p, ctypes.c_int]
    if lib.sph_debug_sub2(sub2.ctypes.data, nwg) == 0:
        ok2 = (sub2[:, 2] >= buf[:, 0]) & (sub2[:, 0] >= sub2[:, 2]) & (sub2[:, 1] >= sub2[:, 0]) & (sub2[:, 1] <= buf[:, 1])
        g = ((sub2[:, 2].astype(np.int64) - t0) / 100.0)[ok2]; a_ = ((sub2[:, 0].astype(np.int64) - t0) / 100.0)[ok2]; b_ = ((sub2[:, 1].astype(np.int64) - t0) / 100.0)[ok2]
        out["staging_head_us_mean"] = {"workgroups": int(ok2.sum()), "gate read (control block)": float(g.mean()), "stage_cnt arrived": float((a_ - g).mean()),
                                       "runs arrived, thread 0 expanded its run": float((b_ - a_).mean())}
# in flight over time
grid = np.arange(0, e.max(), 1.0)
out["in_flight_every_us"] = [int(((b <= t) & (e > t)).sum()) for t in grid]
# the slowest workgroups: who are they?
slow = np.argsort(-life)[:12]
out["slowest"] = [{"blockIdx": int(i), "tile": int(tile[i]), "xcd": int(xcc[i]), "begin_us": float(b[i]), "life_us": float(life[i])} for i in slow]
print(json.dumps(out))
