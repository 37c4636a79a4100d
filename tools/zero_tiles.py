#!/usr/bin/env python3
"""How local is the work of the constant-density loop?  After N steps of dfsph_1m: the share of particles whose stiffness is nonzero in the
last density iteration (rho* > rho0), where they sit, and the share of particles within h / 2h of one of them (the particles whose v* a
correction sweep changes / whose rho* the next residual sweep would have to recompute), per 256-particle tile and per 64-particle wave."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import cKDTree
from cfd_taichi_amd import _native as nat, scenes
sim = nat.Simulation(nat.config_from_dict(scenes.get("dfsph_1m")))
for target in (60, 100, 200):
    while sim.scalar(nat.S_SIMULATE_CNT) < target:
        st = sim.step_dfsph(1)
    ids, ra = sim.download_local(nat.F_RHO_ADV)
    _, pos = sim.download_local(nat.F_POS)
    nz = ra > np.float32(1000.0)
    tree = cKDTree(pos[nz])
    d, _ = tree.query(pos, k=1, distance_upper_bound=0.45)
    n = len(ids)
    out = ["step %d (n_dens %d): nonzero %.4f of the particles" % (target, st.n_dens, nz.mean())]
    for r, what in ((0.1, "within h (v* changes in D7)"), (0.2, "within 2h (rho* changes in the next D6)"), (0.4, "within 4h")):
        near = d <= r
        t256 = near[: n // 256 * 256].reshape(-1, 256).any(1).mean()
        t64 = near[: n // 64 * 64].reshape(-1, 64).any(1).mean()
        out.append("%s: particles %.3f, 64-particle waves %.3f, 256-particle tiles %.3f" % (what, near.mean(), t64, t256))
    yq = np.quantile(pos[nz][:, 1], [0.1, 0.5, 0.9]) if nz.any() else [0, 0, 0]
    out.append("height of the nonzero particles q10/q50/q90 %.2f %.2f %.2f (column %.2f)" % (yq[0], yq[1], yq[2], pos[:, 1].max()))
    print(" | ".join(out), flush=True)
