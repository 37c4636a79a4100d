#!/usr/bin/env python3
"""Health of a long RELAXED run next to the exact one: 1500 steps of dfsph_1m on two handles (exact, relaxed) from rest -- iteration
counts per 100 steps, density error, dt, list lengths, lost particles, finite state, bulk statistics of the two clouds (they are two
different members of the same chaotic ensemble after a few hundred steps: compared through moments, not particle by particle)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cfd_taichi_amd import _native as nat, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "dfsph_1m"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
cfg = scenes.get(scene)
ex = nat.Simulation(nat.config_from_dict(cfg))
rx = nat.Simulation(nat.config_from_dict(cfg, arith=nat.ARITH_RELAXED))
t0 = time.time()
for s in range(0, total, 100):
    nd = [[], []]; mx = [0, 0]; lost = [0, 0]; err = [0.0, 0.0]
    for _ in range(100):
        for k, h in enumerate((ex, rx)):
            st = h.step_dfsph(1)
            nd[k].append(st.n_dens); mx[k] = max(mx[k], st.max_nbrs); lost[k] = max(lost[k], st.lost); err[k] = max(err[k], st.dens_err)
            assert st.capped == 0
    pe, pr = ex.download(nat.F_POS).astype(np.float64), rx.download(nat.F_POS).astype(np.float64)
    ve, vr = ex.download(nat.F_VEL).astype(np.float64), rx.download(nat.F_VEL).astype(np.float64)
    assert np.isfinite(pr).all() and np.isfinite(vr).all()
    ke = [0.5 * (v * v).sum() / len(v) for v in (ve, vr)]
    print("step %4d  n_dens mean exact %.2f relaxed %.2f | max dens_err %.3f %.3f | max_nbrs %d %d lost %d %d | centre of mass x %.5f %.5f y %.5f %.5f | mean kinetic energy %.5f %.5f | front x99.9 %.4f %.4f | %.0f s" % (
        s + 100, np.mean(nd[0]), np.mean(nd[1]), err[0], err[1], mx[0], mx[1], lost[0], lost[1], pe[:, 0].mean(), pr[:, 0].mean(), pe[:, 1].mean(), pr[:, 1].mean(),
        ke[0], ke[1], np.quantile(pe[:, 0], 0.999), np.quantile(pr[:, 0], 0.999), time.time() - t0), flush=True)
