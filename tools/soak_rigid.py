import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from cfd_taichi_amd import _native as nat, mesh, scenes
from oracle import oracle as orc
def run(scene, solver, dt, steps):
    cfg = scenes.get(scene); cfg["solver"]["name"] = solver; cfg["solver"]["delta_time"] = dt
    rg = mesh.rigid_from_config(cfg)
    sim = nat.Simulation(nat.config_from_dict(cfg), rigid=rg); o = orc.Oracle(cfg, solver=solver, num_threads=16, rigid=rg)
    for s in range(steps):
        try:
            st = sim.step(1) if solver != "wcsph" else sim.step_wcsph(1)
        except nat.SphError as e:
            print(scene, solver, "GPU error at step", s, e); break
        {"wcsph": o.step_wcsph, "dfsph": lambda n: o.step_dfsph(n, 100), "pcisph": o.step_pcisph, "iisph": o.step_iisph}[solver](1)
        sim.rigid_step(); o.rigid_step()
        if (s + 1) % 100 == 0:
            a, b = sim.download(nat.F_POS), o.get(orc.F_POS)
            ra, rb = sim.rigid_scalars(), o.rigid_scalars()
            eq = np.array_equal(a, b, equal_nan=True) and np.float32(ra["centroid"]).tolist() == np.float32(rb["centroid"]).tolist() and np.float32(ra["omega"]).tolist() == np.float32(rb["omega"]).tolist()
            print(scene, solver, s + 1, "equal" if eq else "MISMATCH", int((a != b).sum()), ra["centroid"], flush=True)
            if not eq: break
    sim.close(); o.close()
run("dfsph_rigid_small", "dfsph", 1e-3, 700)
run("dfsph_rigid_tilted", "dfsph", 1e-3, 500)
run("dfsph_rigid_small", "iisph", 1e-3, 500)
run("dfsph_rigid_small", "wcsph", 2.5e-4, 1500)
run("dfsph_rigid_small", "pcisph", 2.5e-4, 600)
