import sys, statistics
sys.path.insert(0, "/root/repo")
from cfd_taichi_amd import _native as nat, scenes
sim = nat.Simulation(nat.config_from_dict(scenes.get("dfsph_1m")))
sim.step_dfsph(60); sim.build_neighbors()
for which, label in ((0, "residual"), (1, "correct"), (5, "finalize alone"), (4, "residual+finalize"), (6, "residual+correct"), (7, "residual+finalize+correct")):
    t = [sim.tune_time(which, 0, 20) for _ in range(6)]
    print("%-28s min %.1f median %.1f us per repetition" % (label, min(t), statistics.median(t)))
