"""GPU suite: every scene file the reference ships (config/*.json, default.json), driven the way main.py drives it -- module
`<name>_solver`, class `<name>_solver`, ctor (ps, config) (main.py:64-71), `solver.step()` per frame and `rs.step()` when a body is
active (:165-171) -- against the oracle on the same config dict.  Geometry, solver and dt are the shipped ones (scenes.py restates
the numbers; the camera keys are dropped); the files that carry quirks at key level are here too: wcsph_config_backup.json has no
`boundary_handle` key (default True, solver_base.py:31), default.json and breaking_dam_demo.json call their body block "solid1", so
`config.get('solid', {})` is empty and they are fluid-only scenes with clamp walls.

Shipped file -> scene: breaking_dam_30k.json (names iisph) -> breaking_dam_30k_iisph; breaking_dam_demo.json; coupling_demo.json;
dam_flush_cube.json; dfsph_config_backup.json -> dfsph_small; experiment1_config.json; experiment2_config.json; iisph / pbf / pcisph /
wcsph_config_backup.json; default.json."""
import importlib

import numpy as np
import pytest

from cfd_taichi_amd import ParticleSystem, mesh, rigid_solver, scenes
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

SHIPPED = [  # scene, frames
    ("breaking_dam_30k_iisph", 6), ("breaking_dam_demo", 4), ("coupling_demo", 5), ("dam_flush_cube", 5), ("dfsph_small", 20),
    ("experiment1", 6), ("experiment2", 20), ("iisph_config_backup", 20), ("pbf_small", 20), ("pcisph_config_backup", 20),
    ("wcsph_config_backup", 40), ("default", 6),
]


def oracle_frame(o, name, coupled):
    if name == "wcsph":
        o.step_wcsph(1)
    elif name == "dfsph":
        o.step_dfsph(1, 100)
    elif name == "pcisph":
        o.step_pcisph(1)
    elif name == "iisph":
        o.step_iisph(1)
    else:
        o.step_pbf(1)
    if coupled:
        o.rigid_step()


@pytest.mark.parametrize("scene,frames", SHIPPED)
def test_shipped_config_through_the_reference_frame_loop(scene, frames):
    config = scenes.get(scene)
    name = config["solver"]["name"]
    ps = ParticleSystem(config)
    module = importlib.import_module("cfd_taichi_amd.%s_solver" % name)          # main.py:65-68
    solver = getattr(module, "%s_solver" % name)(ps, config)
    if name in ("dfsph", "pcisph", "iisph"):
        solver.verbose = False
    rs = rigid_solver(ps, config) if config.get("solid", {}) else None           # main.py:69-71
    o = orc.Oracle(config, num_threads=8, rigid=mesh.rigid_from_config(config) if config.get("solid", {}) else None)
    assert ps.particle_num == o.N
    coupled = rs is not None and ps.active_rigid[None] == 1
    assert (rs is not None) == bool(config.get("solid", {}))
    for _ in range(frames):
        for _ in range(config["solver"]["iter_cnt"]):                             # main.py:166-171
            solver.step()
        if coupled:
            for _ in range(config["solver"]["iter_cnt"]):
                rs.step()
        oracle_frame(o, name, coupled)
    pos, ref = ps.fluid_particles.pos.to_numpy(), o.get(orc.F_POS)
    assert pos.shape == ref.shape and np.isfinite(pos).all()
    assert np.abs(pos.astype(np.float64) - ref).max() <= 1e-5 * np.abs(ref).max()             # north_star's bar
    assert np.array_equal(pos, ref), int((pos != ref).any(1).sum())
    assert np.array_equal(ps.fluid_particles.vel.to_numpy(), o.get(orc.F_VEL))
    if name == "dfsph":
        assert solver.delta_time[None] == pytest.approx(o.dt, rel=0, abs=0)
    if coupled:
        assert np.array_equal(ps.rigid_particles.pos.to_numpy(), o.get(orc.F_RIGID_POS))
    o.close()
