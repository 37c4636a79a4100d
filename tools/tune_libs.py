"""Interleaved timing of the DFSPH sweeps for SEVERAL builds of the library inside one process (one clock state):
    tools/tune_libs.py scene advance_steps name=path[:lds[:ENV=value[,ENV2=value2]]] ...      (the ENVs are set while that handle is created)
Each build gets its own handle on the same scene advanced by the same steps; rounds alternate between the builds.
TUNE_COPY_STATE=1: only the first build advances the scene, the others receive its positions / velocities / warm_start_k (for
removal-experiment builds whose results are wrong on purpose)."""
import os
os.environ.setdefault("SPH_DEV", "1")     # tools run with development overrides enabled (sph_overrides reports them)
import random
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfd_taichi_amd import _native as nat, scenes  # noqa: E402

scene, advance = sys.argv[1], int(sys.argv[2])
sims = {}
for spec in sys.argv[3:]:
    name, rest = spec.split("=", 1)
    path, lds, env = (rest.split(":") + ["", ""])[:3]
    nat._lib = None
    os.environ["SPH_LIB"] = os.path.abspath(path)
    envs = [e.split("=") for e in env.split(",") if e]          # ENV=value[,ENV2=value2 ...]
    for k, v in envs:
        os.environ[k] = v
    sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
    for k, _ in envs:
        del os.environ[k]
    if not (os.environ.get("TUNE_COPY_STATE") == "1" and sims):
        sim.step_dfsph(advance)
        state = [sim.download(f) for f in (nat.F_POS, nat.F_VEL, nat.F_WARM_K)]
    else:
        for f, v in zip((nat.F_POS, nat.F_VEL, nat.F_WARM_K), state):
            sim.upload(f, v)
    sim.build_neighbors()
    sims[name] = (sim, int(lds or 0))
for which, label in ((0, "div_residual"), (2, "dens_residual"), (1, "div_correct"), (3, "sort+build_nl")):
    res = {n: [] for n in sims}
    order = list(sims)
    for _ in range(int(os.environ.get("TUNE_ROUNDS", "12"))):
        random.shuffle(order)          # the clock reacts to what ran just before: randomise the order, compare medians
        for n in order:
            sim, lds = sims[n]
            res[n].append(sim.tune_time(which, lds, 10 if which != 3 else 4))
    print(label, {n: (round(min(t), 1), round(statistics.median(t), 1)) for n, t in res.items()})
