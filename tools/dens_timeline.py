#!/usr/bin/env python3
"""Life of every workgroup of ONE iteration of the density loop (ab/libsph_dens_timeline.so from tools/removal_build.py dens_timeline): the residual
sweep D6 and the correction sweep D7 of iteration `it` of step advance + 1 -- who leaves after the need word, who after the per-particle check,
who works, when each starts and ends, how many are in flight.

    python tools/removal_build.py dens_timeline && python tools/dens_timeline.py [scene] [advance_steps] [iteration]"""
import ctypes
import json
import os
os.environ.setdefault("SPH_DEV", "1")
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SPH_LIB"] = os.path.join(ROOT, "ab", "libsph_dens_timeline.so")
from cfd_taichi_amd import _native as nat, scenes  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "dfsph_1m"
advance = int(sys.argv[2]) if len(sys.argv) > 2 else 60
it = int(sys.argv[3]) if len(sys.argv) > 3 else 6
sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
sim.step_dfsph(advance)
lib = nat.load()
lib.sph_debug_flow_stamp.argtypes = [ctypes.c_void_p]
lib.sph_debug_dens_timeline.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
s0 = lib.sph_debug_flow_stamp(sim._h)
assert lib.sph_debug_dens_capture(s0 + 2 * it - 1, s0 + 2 * it) == 0        # the step's sweeps take stamps s0 + 1 (D6 of iteration 1), s0 + 2 (D7), ...
st = sim.step_dfsph(1)
assert st.n_dens >= it, (st.n_dens, it)
nwg = (sim.n_fluid + 255) // 256 + 1
out = {"scene": scene, "step": advance + 1, "iteration": it, "n_dens": st.n_dens, "tiles": nwg - 1}
for which, name in ((6, "D6 residual"), (7, "D7 correction")):
    buf = np.zeros((nwg, 4), dtype=np.uint64)
    assert lib.sph_debug_dens_timeline(which, buf.ctypes.data, nwg) == 0
    ok = buf[:, 1] > 0
    t0, t1 = buf[ok, 0].astype(np.int64), buf[ok, 1].astype(np.int64)
    outcome = (buf[ok, 2] & np.uint64(0xff)).astype(np.int64)
    blockidx = np.nonzero(ok)[0]
    base = t0.min()
    b, e = (t0 - base) / 100.0, (t1 - base) / 100.0          # wall_clock64: 100 MHz -> us
    life = e - b
    rec = {"workgroups_recorded": int(ok.sum()), "span_us": float(e.max())}
    for o, what in ((0, "left after the need word"), (1, "left after the per-particle check"), (2, "worked")):
        m = outcome == o
        if m.any():
            rec[what] = {"workgroups": int(m.sum()), "life_us_mean": float(life[m].mean()), "life_us_p90": float(np.percentile(life[m], 90)),
                         "first_begin_us": float(b[m].min()), "last_begin_us": float(b[m].max()), "begin_us_p50": float(np.percentile(b[m], 50)),
                         "begin_us_p90": float(np.percentile(b[m], 90)), "last_end_us": float(e[m].max()),
                         "blockIdx_p50": int(np.percentile(blockidx[m], 50)), "blockIdx_max": int(blockidx[m].max())}
    grid = np.arange(0, e.max(), 1.0)
    rec["in_flight_every_us"] = [int(((b <= t) & (e > t)).sum()) for t in grid]
    rec["working_in_flight_every_us"] = [int(((b <= t) & (e > t) & (outcome == 2)).sum()) for t in grid]
    w = outcome == 2
    extra = (buf[ok, 2] >> np.uint64(32)).astype(np.int64)
    extra = np.where(extra >= 2 ** 31, extra - 2 ** 32, extra)          # stage_cnt of a working tile: particles | runs << 16 | kStageLists16, or -1 = not staged
    tile = (buf[ok, 3] & np.uint64(0xffff)).astype(np.int64)
    ph = [((buf[ok, 3] >> np.uint64(16 * (k + 1))) & np.uint64(0xffff)).astype(np.float64) / 100.0 for k in range(3)]      # staged, pairs done, walls done (us after begin)
    late = np.argsort(-e)[:12]
    rec["last_to_end"] = [{"blockIdx": int(blockidx[i]), "tile": int(tile[i]), "outcome": int(outcome[i]), "begin_us": float(b[i]), "life_us": float(life[i]),
                           "staged_at": float(ph[0][i]), "pairs_done_at": float(ph[1][i]), "walls_done_at": float(ph[2][i]), "staged_particles": int(extra[i] & 0xffff) if extra[i] >= 0 else -1, "cell_runs": int((extra[i] >> 16) & 0x3fff) if extra[i] >= 0 else -1} for i in late]
    if w.any():
        rec["worked: phases_us_mean"] = {"check + staging": float(ph[0][w].mean()), "fluid pair loop": float((ph[1] - ph[0])[w].mean()), "wall terms": float((ph[2] - ph[1])[w].mean()),
                                         "epilogue": float((life - ph[2])[w].mean())}
        rec["worked: phases_us_p99"] = {"check + staging": float(np.percentile(ph[0][w], 99)), "fluid pair loop": float(np.percentile((ph[1] - ph[0])[w], 99)),
                                        "wall terms": float(np.percentile((ph[2] - ph[1])[w], 99)), "epilogue": float(np.percentile((life - ph[2])[w], 99))}
        unst = w & (extra < 0)
        rec["worked: not staged"] = {"workgroups": int(unst.sum()), "life_us_mean": float(life[unst].mean()) if unst.any() else None}
        st_ = w & (extra >= 0)
        nst = (extra & 0xffff)
        rec["worked: staged"] = {"workgroups": int(st_.sum()), "life_us_mean": float(life[st_].mean()), "life_vs_staged_particles_corr": float(np.corrcoef(nst[st_], life[st_])[0, 1]),
                                 "staged_particles_mean": float(nst[st_].mean()), "life_us_by_quartile_of_staged_particles": [float(life[st_][(nst[st_] >= lo) & (nst[st_] <= hi)].mean())
                                                                                                                            for lo, hi in zip(np.percentile(nst[st_], [0, 25, 50, 75]), np.percentile(nst[st_], [25, 50, 75, 100]))]}
    out[name] = rec
print(json.dumps(out))
