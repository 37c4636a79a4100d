"""Two builds of the library advance the same scene in lock step and must stay bit-identical:  tools/soak_libs.py scene steps every libA.so libB.so
(e.g. the default build against ab/libsph_bnl_notable.so from tools/removal_build.py: k_build_nl's per-wave cell tables against the per-lane path, over a whole collapse)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes
scene, steps, every = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = scenes.get(scene)
sims = [nat.Simulation(nat.config_from_dict(cfg), lib=nat.bind_core(os.path.abspath(p))) for p in sys.argv[4:6]]
wc = cfg["solver"]["name"] == "wcsph"
t0 = time.time()
done = 0
while done < steps:
    n = min(every, steps - done)
    if wc:
        for s in sims:
            s.step_wcsph(n)
    else:
        for _ in range(n):
            a, b = sims[0].step(1), sims[1].step(1)
            assert (a.n_div, a.n_dens, a.div_err, a.dens_err, a.dt, a.max_nbrs, a.lost) == (b.n_div, b.n_dens, b.div_err, b.dens_err, b.dt, b.max_nbrs, b.lost), done
    done += n
    for f in (nat.F_POS, nat.F_VEL, nat.F_RHO):
        assert np.array_equal(sims[0].download(f), sims[1].download(f), equal_nan=True), (done, f)
    print("step %d: identical (%.0f s)" % (done, time.time() - t0), flush=True)
