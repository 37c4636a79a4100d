"""How many particles would a 256-particle workgroup have to stage (all particles of the 27-cell neighbourhoods of its own particles)?
tools/stage_footprint.py scene steps  ->  distribution over workgroups, for the device order in use (SPH_CELL_ORDER)."""
import os
os.environ.setdefault("SPH_DEV", "1")     # tools run with development overrides enabled (sph_overrides reports them)
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes  # noqa: E402

scene, steps = sys.argv[1], int(sys.argv[2])
block = int(os.environ.get("STAGE_BLOCK", "256"))
cfg = scenes.get(scene)
sim = nat.Simulation(nat.config_from_dict(cfg))
sim.step(steps)
sim.build_neighbors()
ids, pos = sim.download_local(nat.F_POS)
h = np.float32(4 * cfg["scene"]["particle_radius"])
c3 = np.floor(pos / h).astype(np.int64)
gx, gy, gz = sim.grid
cid = c3[:, 0] + c3[:, 1] * gx * gz + c3[:, 2] * gx
count = np.bincount(cid, minlength=gx * gy * gz)
n = len(cid)
res, cells = [], []
rng = np.random.default_rng(0)
blocks = rng.choice((n + block - 1) // block, size=min(400, (n + block - 1) // block), replace=False)
off = np.array([dx + dy * gx * gz + dz * gx for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)])
for b in blocks:
    home = np.unique(cid[b * block:(b + 1) * block])
    nb = np.unique((home[:, None] + off[None, :]).ravel())
    nb = nb[(nb >= 0) & (nb < len(count))]
    res.append(int(count[nb].sum())); cells.append(len(nb))
res, cells = np.array(res), np.array(cells)
print(scene, "steps", steps, "order", os.environ.get("SPH_CELL_ORDER", "auto"), "block", block)
print("  staged particles per block: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %d" % (res.mean(), np.percentile(res, 50), np.percentile(res, 90), np.percentile(res, 99), res.max()))
print("  neighbourhood cells per block: mean %.0f max %d;  particles per occupied cell: %.1f" % (cells.mean(), cells.max(), n / (count > 0).sum()))
