"""Do identical handles run at the same speed?  tools/placement.py scene advance n_handles
Creates n handles of the same scene in one process and times the DFSPH sweeps on each, interleaved in random order."""
import os
import random
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes  # noqa: E402

scene, advance, nh = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sims = []
for k in range(nh):
    sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
    sim.step_dfsph(advance)
    sim.build_neighbors()
    sims.append(sim)
for which, label in ((0, "div_residual"), (1, "div_correct"), (3, "sort+build_nl")):
    res = [[] for _ in sims]
    order = list(range(nh))
    for _ in range(12):
        random.shuffle(order)
        for k in order:
            res[k].append(sims[k].tune_time(which, 0, 10 if which != 3 else 4))
    print(label, [round(statistics.median(t), 1) for t in res])
