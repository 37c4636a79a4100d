import os, sys, copy
sys.path.insert(0, os.getcwd())
from cfd_taichi_amd import _native as nat, mesh, scenes
for couple in (True, False):
    cfg = copy.deepcopy(scenes.get("dfsph_rigid_2m_clear"))
    cfg["solver"]["fs_couple"] = couple
    rg = mesh.rigid_from_config(cfg)
    sim = nat.Simulation(nat.config_from_dict(cfg), rigid=rg)
    for _ in range(60):
        sim.step_dfsph(1); sim.rigid_step()
    sim.profile_enable(True); sim.profile_reset()
    for _ in range(20):
        sim.step_dfsph(1); sim.rigid_step()
    pr = sim.profile()
    print("fs_couple", couple, {k: round(v[0] / v[1] * 1e3, 1) for k, v in pr.items() if k in ("dfsph_div_residual", "dfsph_div_correct", "dfsph_ext_force", "dfsph_density_alpha", "build_nl", "dfsph_warm_start")})
    sim.close()
