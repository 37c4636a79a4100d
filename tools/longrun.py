import os, sys, time
sys.path.insert(0, os.getcwd())
from cfd_taichi_amd import _native as nat, scenes
for scene, n in (("dfsph_1m", 2000), ("wcsph_250k", 6000)):
    sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
    t0 = time.time(); mx = 0; mxw = 0; nd = []
    for s in range(0, n, 100):
        if scene.startswith("dfsph"):
            for _ in range(100):
                st = sim.step_dfsph(1); mx = max(mx, st.max_nbrs); mxw = max(mxw, st.max_wall_nbrs); nd.append(st.n_dens)
            print(scene, s + 100, "max_nbrs", mx, mxw, "n_dens last100 mean", sum(nd[-100:]) / 100, "capped", st.capped, "lost", st.lost, "dt", st.dt, round(time.time() - t0, 1), flush=True)
        else:
            sim.step_wcsph(100); sim.synchronize()
            print(scene, s + 100, round(time.time() - t0, 1), flush=True)
    sim.close()
