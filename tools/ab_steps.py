"""Interleaved timing of WHOLE steps for several builds / settings of the library inside one process (one clock state):
    tools/ab_steps.py scene warm_steps name=path[:ENV=value] ...
Every variant gets its own handle on the same scene; all advance in lock step (results are bit-identical across variants, so the
state they time is the same), rounds of `AB_CHUNK` steps are timed in random order.  AB_SOLVER overrides the scene's solver."""
import os
os.environ.setdefault("SPH_DEV", "1")     # tools run with development overrides enabled (sph_overrides reports them)
import random
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes  # noqa: E402

scene, warm = sys.argv[1], int(sys.argv[2])
chunk, rounds = int(os.environ.get("AB_CHUNK", "20")), int(os.environ.get("AB_ROUNDS", "10"))
sims = {}
for spec in sys.argv[3:]:
    name, rest = spec.split("=", 1)
    path, env = (rest.split(":") + [""])[:2]
    nat._lib = None
    os.environ["SPH_LIB"] = os.path.abspath(path)
    if env:
        os.environ[env.split("=")[0]] = env.split("=")[1]
    sims[name] = nat.Simulation(nat.config_from_dict(scenes.get(scene), solver_name=os.environ.get("AB_SOLVER")))
    if env:
        del os.environ[env.split("=")[0]]
    sims[name].step(warm)
    sims[name].synchronize()
res = {n: [] for n in sims}
order = list(sims)
for _ in range(rounds):
    random.shuffle(order)
    for n in order:
        t0 = time.perf_counter()
        sims[n].step(chunk)
        sims[n].synchronize()
        res[n].append((time.perf_counter() - t0) / chunk * 1e3)
n_fluid = next(iter(sims.values())).n_fluid
for n, t in res.items():
    print("%-10s ms/step min %.3f median %.3f  -> %.1f Mparticle-steps/s (median)" % (n, min(t), statistics.median(t), n_fluid / statistics.median(t) / 1e3))
