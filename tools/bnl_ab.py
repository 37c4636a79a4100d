#!/usr/bin/env python3
"""sort + list build of one state under several builds of the library, interleaved:  python tools/bnl_ab.py scene advance name=lib.so ...
(the first library advances the scene, the others receive its state; mean of `reps` timings by HIP events, sph_tune_time(3))"""
import os
os.environ.setdefault("SPH_DEV", "1")
import subprocess
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import numpy as np
    from cfd_taichi_amd import _native as nat, scenes
    scene, advance, state = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    sim = nat.Simulation(nat.config_from_dict(scenes.get(scene)))
    if os.path.exists(state):
        z = np.load(state)
        sim.upload(nat.F_POS, z["pos"]); sim.upload(nat.F_VEL, z["vel"]); sim.upload(nat.F_WARM_K, z["warm"]); sim.set_dt(float(z["dt"]))
    else:
        sim.step_dfsph(advance)
        np.savez(state, pos=sim.download(nat.F_POS), vel=sim.download(nat.F_VEL), warm=sim.download(nat.F_WARM_K), dt=sim.scalar(nat.S_DELTA_TIME))
    sim.step_dfsph(1)
    t = [sim.tune_time(3, 0, 1) for _ in range(12)][2:]
    print(json.dumps({"mean_us": sum(t) / len(t), "min_us": min(t)}))
    sys.exit(0)
scene, advance = sys.argv[1], sys.argv[2]
libs = [a.split("=", 1) for a in sys.argv[3:]]
state = "/tmp/bnl_ab_state_%s_%s.npz" % (scene, advance)
if os.path.exists(state):
    os.remove(state)
for rnd in range(2):
    for name, lib in libs:
        env = dict(os.environ, SPH_LIB=os.path.join(ROOT, lib))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", scene, advance, state], env=env, capture_output=True, text=True)
        print(name, out.stdout.strip() or out.stderr[-400:], flush=True)
