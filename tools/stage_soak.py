"""Long staged-vs-unstaged comparison on large scenes: tools/stage_soak.py scene steps [check_every]
Two handles (SPH_STAGE=1 / 0) advance in lock step; positions, velocities and the step statistics must stay bit-identical while the
dam collapses (workgroups drift between staged and unstaged as the flow thins out)."""
import os, sys, time
os.environ.setdefault("SPH_DEV", "1")     # tools run with development overrides enabled (sph_overrides reports them)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes
scene, steps = sys.argv[1], int(sys.argv[2])
every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
sims = []
for stage in ("1", "0"):
    os.environ["SPH_STAGE"] = stage
    sims.append(nat.Simulation(nat.config_from_dict(scenes.get(scene))))
del os.environ["SPH_STAGE"]
t0 = time.time()
for s in range(steps):
    a, b = sims[0].step(1), sims[1].step(1)
    assert (a.n_div, a.n_dens, a.div_err, a.dens_err, a.dt, a.max_nbrs, a.lost) == (b.n_div, b.n_dens, b.div_err, b.dens_err, b.dt, b.max_nbrs, b.lost), s   # all four solvers fill SphStepStats except wcsph
    if (s + 1) % every == 0 or s + 1 == steps:
        for f in (nat.F_POS, nat.F_VEL):
            assert np.array_equal(sims[0].download(f), sims[1].download(f), equal_nan=True), (s, f)
        print("step %d: identical; n_dens %d max_nbrs %d lost %d dt %.3g (%.0f s)" % (s + 1, a.n_dens, a.max_nbrs, a.lost, a.dt, time.time() - t0), flush=True)
