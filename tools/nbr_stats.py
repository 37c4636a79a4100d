"""Neighbour-count statistics of a scene after N steps (pairs per sweep, lane utilisation of the wave-level list walk):
    tools/nbr_stats.py scene steps"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes
sim = nat.Simulation(nat.config_from_dict(scenes.get(sys.argv[1])))
sim.step(int(sys.argv[2]))
sim.build_neighbors()
ids, _ = sim.download_local(nat.F_POS)
cnt = sim.download(nat.F_NBR_COUNT).astype(np.int64)[ids]          # device order
pad = (-len(cnt)) % 64
w = np.pad(cnt, (0, pad)).reshape(-1, 64)
print({"particles": len(cnt), "pairs": int(cnt.sum()), "mean": float(cnt.mean()), "max": int(cnt.max()),
       "wave_max_mean": float(w.max(1).mean()), "lane_utilisation": float(cnt.sum() / (w.max(1).sum() * 64)),
       "groups_of_4_per_wave_mean": float(np.ceil(w.max(1) / 4).mean()), "groups_of_8_per_wave_mean": float(np.ceil(w.max(1) / 8).mean()),
       "hist": np.bincount(np.minimum(cnt, 70) // 10).tolist()})

# what a count-sorted lane assignment inside each workgroup of 256 particles (4 waves) would give
pad = (-len(cnt)) % 256
g = np.sort(np.pad(cnt, (0, pad)).reshape(-1, 256), axis=1).reshape(-1, 64)
print({"sorted_within_256": {"wave_max_mean": float(g.max(1).mean()), "lane_utilisation": float(cnt.sum() / (g.max(1).sum() * 64)),
                             "groups_of_8_per_wave_mean": float(np.ceil(g.max(1) / 8).mean()),
                             "groups_of_8_unsorted": float(np.ceil(w.max(1) / 8).mean())}})
