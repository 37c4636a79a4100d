import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from cfd_taichi_amd import _native as nat, scenes
def run(scene, steps, stage):
    os.environ["SPH_STAGE"] = "1" if stage else "0"
    os.environ["SPH_CELL_ORDER"] = "morton"
    sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
    st = [sim.step(1) for _ in range(steps)]
    out = (sim.download(nat.F_POS), sim.download(nat.F_VEL), [(s.n_div, s.n_dens, s.div_err, s.dens_err) for s in st])
    sim.close()
    return out
for scene, steps in (("dfsph_small", 30), ("dfsph_tiny_wall", 40), ("breaking_dam_30k_dfsph", 10), ("dfsph_1m", 6)):
    a, b = run(scene, steps, True), run(scene, steps, False)
    print(scene, "pos", np.array_equal(a[0], b[0]), "vel", np.array_equal(a[1], b[1]), "stats", a[2] == b[2], flush=True)
