"""Generates golden vectors FROM THE REFERENCE ITSELF -- the day a taichi wheel is present in the build container.

    python tests/golden/make_reference_fixtures.py [/root/reference]

It imports the reference's own ParticleSystem and solver classes UNCHANGED from the given directory (nothing of them is copied here, and
nothing of them ever travels: only the arrays written below do), runs a scene on ti.cpu, and writes
    tests/golden/ref_<scene>_<solver>.npz      pos / vel after each of the steps listed in STEPS, delta_time, particle counts
which tests/test_oracle_kats.py::test_oracle_against_reference_fixtures compares with the oracle when the files exist.

Status in this pipeline: `import taichi` raises ModuleNotFoundError in the build container (taichi==1.6.0, requirements.txt:2, is not
installed and there is no network), so this script has never run and the oracle stays "parity unpinned" (DESIGN.md section 2).  The
reference's cell lists are appended from a parallel loop (ParticleSystem.py:388-397), so even with fixtures only WCSPH can be compared
at 1e-5 beyond a handful of steps ("the reference's own nondeterminism"); the fixtures therefore also store a second
run of the same scene, so that the comparison can be held to the reference's own run-to-run spread.
"""
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
STEPS = (1, 2, 5, 10, 20)
SCENES = {         # name -> (config file of the reference, solver.name override)
    "dfsph_config_backup": ("config/dfsph_config_backup.json", "dfsph"),
    "wcsph_config_backup": ("config/wcsph_config_backup.json", "wcsph"),
}


def run(ref_dir, config_path, solver_name, ti):
    with open(os.path.join(ref_dir, config_path)) as f:
        config = json.load(f)
    config["solver"]["name"] = solver_name
    config.pop("solid", None)
    particle_system = importlib.import_module("ParticleSystem")
    ps = particle_system.ParticleSystem(config)
    module = importlib.import_module(solver_name + "_solver")
    solver = getattr(module, solver_name + "_solver")(ps, config)
    out = {"n_fluid": np.int64(ps.particle_num), "n_wall": np.int64(ps.boundary_particles_num)}
    for step in range(1, max(STEPS) + 1):
        solver.step()
        if step in STEPS:
            out["pos_%d" % step] = ps.fluid_particles.pos.to_numpy().astype(np.float32)
            out["vel_%d" % step] = ps.fluid_particles.vel.to_numpy().astype(np.float32)
            out["dt_%d" % step] = np.float32(solver.delta_time[None])
    return out


def main():
    ref_dir = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    try:
        import taichi as ti
    except ModuleNotFoundError as e:
        raise SystemExit("taichi is not installed (%s): the reference cannot run here, no fixtures written" % e)
    sys.path.insert(0, ref_dir)
    os.chdir(ref_dir)
    for name, (config_path, solver_name) in SCENES.items():
        runs = []
        for _ in range(2):          # twice: the reference's run-to-run spread is part of the fixture
            ti.init(arch=ti.cpu)    # main.py:22 (the reference's own default is ti.gpu; configs[0] of BASELINE.json names ti.cpu)
            for mod in [m for m in list(sys.modules) if m in ("ParticleSystem", "solver_base") or m.endswith("_solver")]:
                del sys.modules[mod]
            runs.append(run(ref_dir, config_path, solver_name, ti))
        out = dict(runs[0])
        out.update({"again_" + k: v for k, v in runs[1].items() if k.startswith(("pos_", "vel_"))})
        path = os.path.join(HERE, "ref_%s_%s.npz" % (name, solver_name))
        np.savez_compressed(path, **out)
        print(path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.startswith("again_")})


if __name__ == "__main__":
    main()
