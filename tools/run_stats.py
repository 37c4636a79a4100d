"""Runs of equal cell per workgroup (256 particles) and per wave of the sorted order after N steps -- sizes the per-run tables of
k_build_nl:   tools/run_stats.py scene steps"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes
sim = nat.Simulation(nat.config_from_dict(scenes.get(sys.argv[1])))
sim.step(int(sys.argv[2]))
sim.build_neighbors()
_, pos = sim.download_local(nat.F_POS)
cell = np.floor(pos / np.float32(0.1)).astype(np.int64)
key = (cell[:, 0] * 4096 + cell[:, 1]) * 4096 + cell[:, 2]
for width in (64, 256):
    pad = (-len(key)) % width
    k = np.pad(key, (0, pad), constant_values=-1).reshape(-1, width)
    runs = 1 + (k[:, 1:] != k[:, :-1]).sum(1)
    print(width, {"mean": float(runs.mean()), "p50": int(np.percentile(runs, 50)), "p90": int(np.percentile(runs, 90)), "p99": int(np.percentile(runs, 99)),
                  "p999": int(np.percentile(runs, 99.9)), "max": int(runs.max()),
                  "share_over_32": float((runs > 32).mean()), "share_over_48": float((runs > 48).mean()), "share_over_64": float((runs > 64).mean())})
