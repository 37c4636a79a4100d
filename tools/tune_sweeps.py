"""Interleaved A/B timing of the DFSPH sweeps inside ONE process (one clock state): tools/tune_sweeps.py [scene] [advance_steps]
Prints, per dynamic-LDS setting (= cap on resident waves per CU), the min / median microseconds over rounds."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "dfsph_1m"
advance = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lds_values = [int(v) for v in os.environ.get("TUNE_LDS", "0,16384,24576,32768,40960,49152").split(",")]
sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
sim.step_dfsph(advance)
sim.build_neighbors()
for which, name in ((0, "div_residual"), (2, "dens_residual"), (1, "div_correct")):
    res = {v: [] for v in lds_values}
    for _ in range(6):
        for v in lds_values:
            res[v].append(sim.tune_time(which, v, 10))
    print(name, {v: (round(min(t), 1), round(statistics.median(t), 1)) for v, t in res.items()})
