import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from cfd_taichi_amd import _native as nat, scenes
cfg = scenes.get("dfsph_10m")
cfg["scene"]["box_max"] = [110.0, 8.0, 20.2]
cfg["fluid"]["water_size"] = [100.0, 6.25, 20.0]
t0 = time.time()
sim = nat.Simulation(nat.config_from_dict(cfg))
print("N", sim.n_fluid, "Nb", sim.n_wall, "grid", sim.grid, "create s", round(time.time() - t0, 1), flush=True)
for s in range(4):
    t1 = time.time(); st = sim.step_dfsph(1); sim.synchronize()
    print("step", s, round((time.time() - t1) * 1e3, 1), "ms", st.n_div, st.n_dens, st.max_nbrs, st.lost, st.dt, flush=True)
pos = sim.download(nat.F_POS)
print("finite", bool(np.isfinite(pos).all()), float(pos.min()), pos.max(0), "Mps/s last", round(sim.n_fluid / (time.time() - t1) / 1e6, 1))
rho = sim.download(nat.F_RHO); print("rho", float(rho.min()), float(rho.max()), float(rho.mean()))
