#!/usr/bin/env python3
"""Long parity runs against the oracle, step by step (state compared every `check` steps, iteration counts and residuals every step):
    tools/soak_oracle.py scene steps [check]          SPH_CELL_ORDER=morton puts a small scene on the staged path (LDS staging, 16-bit lists,
                                                     change propagation, wall-gradient cache) that the large scenes run."""
import os
os.environ.setdefault("SPH_DEV", "1")     # tools run with development overrides enabled (sph_overrides reports them)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, mesh, scenes  # noqa: E402
from oracle import oracle as orc  # noqa: E402

scene, steps = sys.argv[1], int(sys.argv[2])
check = int(sys.argv[3]) if len(sys.argv) > 3 else 50
cfg = scenes.get(scene)
solver = cfg["solver"]["name"]
rg = mesh.rigid_from_config(cfg) if cfg.get("solid") else None       # a coupled body (rigid_solver.py): stepped on both sides
sim = nat.Simulation(nat.config_from_dict(cfg), rigid=rg)
def cores():                  # the cgroup's share, not the host's CPU list (an oversubscribed OpenMP team crawls)
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:
        pass
    return n


o = orc.Oracle(cfg, solver=solver, num_threads=cores(), rigid=rg)
t0 = time.time()
iters = []
for s in range(1, steps + 1):
    if solver == "dfsph":
        st = sim.step_dfsph(1); o.step_dfsph(1, 100); so = o.last_stats
        assert (st.n_div, st.n_dens, st.div_err, st.dens_err, st.dt) == (so.n_div, so.n_dens, so.div_err, so.dens_err, so.dt), (s, st.n_div, so.n_div, st.n_dens, so.n_dens)
        iters.append(st.n_dens)
    elif solver in ("pcisph", "iisph"):
        st = (sim.step_pcisph if solver == "pcisph" else sim.step_iisph)(1)
        (o.step_pcisph if solver == "pcisph" else o.step_iisph)(1)
        assert (st.n_dens, st.dens_err) == (o.last_stats.n_dens, o.last_stats.dens_err), (s, st.n_dens, o.last_stats.n_dens)
        iters.append(st.n_dens)
    elif solver == "pbf":
        sim.step_pbf(1); o.step_pbf(1)
    else:
        sim.step_wcsph(1); o.step_wcsph(1)
    if rg is not None:
        fa, fb = sim.download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID), o.get(orc.F_RIGID_FORCE)
        assert np.array_equal(fa, fb), "step %d: force on the body differs" % s
        sim.rigid_step(); o.rigid_step()
        ra, rb = sim.rigid_scalars(), o.rigid_scalars()
        for k in ("centroid", "omega", "vel"):
            assert np.array_equal(np.float32(ra[k]), np.float32(rb[k])), (s, k, ra[k], rb[k])
    if s % check == 0 or s == steps:
        for f, g in ((nat.F_POS, orc.F_POS), (nat.F_VEL, orc.F_VEL)):
            a, b = sim.download(f), o.get(g)
            assert np.array_equal(a, b), "step %d field %d: %d entries differ" % (s, f, int((a != b).sum()))
        print("%s step %d: bit-equal (%.0f s)%s" % (scene, s, time.time() - t0, "; n_dens so far min %d max %d" % (min(iters), max(iters)) if iters else ""), flush=True)
print("%s: %d steps bit-equal to the oracle, order %s" % (scene, steps, os.environ.get("SPH_CELL_ORDER", "auto")))
