"""tools/placement_pmc.py scene advance n_handles: like placement.py but every handle runs its residual sweeps back to back
(no interleaving), so that a rocprofv3 --pmc pass over this script can attribute dispatches to handles by order."""
import os
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
for rnd in range(2):
    print("round", rnd, [round(s.tune_time(0, 0, 10), 1) for s in sims], flush=True)
