import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from cfd_taichi_amd import _native as nat, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "pcisph_1m"
sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
step = sim.step_pcisph if "pcisph" in scene else sim.step_iisph
for target in (30, 60, 100):
    while sim.scalar(nat.S_SIMULATE_CNT) < target:
        st = step(1)
    pr = sim.download(nat.F_PRESS_ITER)
    pos = sim.download(nat.F_POS)
    ids, lpos = sim.download_local(nat.F_POS)
    prl = pr[ids]                      # device (sorted) order
    nz = prl != 0
    n = len(ids)
    t256 = nz[: n // 256 * 256].reshape(-1, 256).any(1).mean()
    t64 = nz[: n // 64 * 64].reshape(-1, 64).any(1).mean()
    yq = np.quantile(pos[pr != 0][:, 1], [0.1, 0.5, 0.9]) if (pr != 0).any() else [0, 0, 0]
    print("step %d iters %d: particles with pressure != 0: %.4f, waves %.3f, tiles %.3f; their height q10/q50/q90 %.2f %.2f %.2f (column %.2f)" % (
        target, st.n_dens, nz.mean(), t64, t256, yq[0], yq[1], yq[2], pos[:, 1].max()))
