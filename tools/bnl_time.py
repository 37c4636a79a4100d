"""Times the sort + neighbour-list build alone (HIP events) for one build of the library: tools/bnl_time.py [scene]."""
import os
os.environ.setdefault("SPH_DEV", "1")     # tools run with development overrides enabled (sph_overrides reports them)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes  # noqa: E402

cfg = scenes.get(sys.argv[1] if len(sys.argv) > 1 else "dfsph_1m")
sim = nat.Simulation(nat.config_from_dict(cfg, solver_name="wcsph"))
sim.profile_enable(True)
for _ in range(8):
    try:
        sim.build_neighbors()
    except nat.SphError:
        pass
sim.synchronize()
print(os.environ.get("SPH_LIB", "default").split("/")[-1],
      {k: (round(v[0] / max(v[1], 1) * 1000, 1), v[1]) for k, v in sim.profile().items() if k in ("build_nl", "hash_count", "order_gather", "scatter", "scan")})
