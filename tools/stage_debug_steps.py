import os, sys
os.environ.setdefault("SPH_DEV", "1")     # tools run with development overrides enabled (sph_overrides reports them)
sys.path.insert(0, os.getcwd())
from cfd_taichi_amd import _native as nat, scenes
sim = nat.Simulation(nat.config_from_dict(scenes.get("dfsph_1m")))
for k in range(170):
    if k in (0, 55, 100, 169): os.environ["SPH_STAGE_DEBUG"] = "1"
    else: os.environ.pop("SPH_STAGE_DEBUG", None)
    st = sim.step_dfsph(1)
    if k in (0, 55, 100, 169): print("step", k + 1, st.n_div, st.n_dens, flush=True)
