"""The PCIe-inclusive rate of the reference's frame loop: every frame one solver step and `ps.fluid_particles.pos.to_numpy()` (what main.py's PLY
export reads, main.py:189-195) -- against the resident rate of the same steps.   tools/pcie_rate.py [scene] [pre-roll] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes
scene = sys.argv[1] if len(sys.argv) > 1 else "dfsph_1m"
pre, steps = int(sys.argv[2]) if len(sys.argv) > 2 else 70, int(sys.argv[3]) if len(sys.argv) > 3 else 60
sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
sim.step(pre)
sim.synchronize()
out = {}
for name, fetch in (("resident", False), ("with_positions_to_host_every_step", True), ("resident_again", False)):
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step(1)
        if fetch:
            pos = sim.download(nat.F_POS)
    sim.synchronize()
    dt = time.perf_counter() - t0
    out[name] = {"Mparticle_steps_per_s": round(sim.n_fluid * steps / dt / 1e6, 1), "ms_per_step": round(dt / steps * 1e3, 3)}
t0 = time.perf_counter()
for _ in range(10):
    pos = sim.download(nat.F_POS)
out["download_pos_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
out["bytes"] = int(pos.nbytes)
print(scene, out)
