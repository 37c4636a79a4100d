"""Two settings of one environment knob give the same bits: tools/stage_check.py KNOB [scene steps ...]"""
import os, sys
os.environ.setdefault("SPH_DEV", "1")     # tools run with development overrides enabled (sph_overrides reports them)
import numpy as np
sys.path.insert(0, os.getcwd())
from cfd_taichi_amd import _native as nat, scenes
knob = sys.argv[1] if len(sys.argv) > 1 else "SPH_STAGE"
def run(scene, steps, on):
    os.environ[knob] = "1" if on else "0"
    os.environ["SPH_CELL_ORDER"] = "morton"
    sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
    st = [sim.step(1) for _ in range(steps)]
    out = (sim.download(nat.F_POS), sim.download(nat.F_VEL), [(s.n_div, s.n_dens, s.div_err, s.dens_err, s.n_div_evals) for s in st])
    sim.close()
    return out
for scene, steps in (("dfsph_small", 30), ("dfsph_tiny_wall", 40), ("breaking_dam_30k_dfsph", 10), ("dfsph_1m", 6)):
    a, b = run(scene, steps, True), run(scene, steps, False)
    print(scene, knob, "pos", np.array_equal(a[0], b[0]), "vel", np.array_equal(a[1], b[1]), "stats", a[2] == b[2], flush=True)
