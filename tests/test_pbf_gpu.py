"""GPU suite, SURVEY.md 8f.4: the PBF step (pbf_solver.py:176-187) against the oracle, bit for bit.

The reference file is stale at the surveyed commit (fluid callbacks written for particle indices, for_all_neighbor passes structs) and
its update_all_pos races on pos / vel; library and oracle read it the same documented way (csrc/sph_pbf_kernels.h): callbacks on the
structs' fields, update_all_pos under the barrier-synchronised schedule.  Parity is therefore "equal to the restatement under that
reading", unpinned like the rest."""
import numpy as np
import pytest

from cfd_taichi_amd import _native as nat
from cfd_taichi_amd import scenes
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        scale = max(float(np.abs(b).max()), 1e-30)
        raise AssertionError("%s differs at %d of %d entries, rel err %.3e, first %s: %r vs %r" % (
            what, len(bad), a.size, float(np.abs(a.astype(np.float64) - b).max()) / scale, bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))


def squeeze(sim, o, factor=0.9):
    """The rest lattice is under-dense by the reference's density sum (no self term): the constraint max(rho / rho_0 - 1, 0) would stay
    at zero until the column has collapsed.  Squeezing the lattice makes lambda and delta_pos act from the first step."""
    pos = o.get(orc.F_POS)
    about = pos.min(0)
    sq = (about + (pos - about) * np.float32(factor)).astype(np.float32)
    o.set(orc.F_POS, sq); sim.upload(nat.F_POS, sq)


@pytest.mark.parametrize("scene,steps,squeezed", [("pbf_tiny_wall", 150, True), ("pbf_tiny_clamp", 150, True), ("pbf_small", 60, False),
                                                  ("pbf_tiny_wall", 400, False)])
def test_pbf_steps(scene, steps, squeezed):
    cfg = scenes.get(scene)
    sim = nat.Simulation(nat.config_from_dict(cfg))
    o = orc.Oracle(cfg, num_threads=8)
    assert (sim.n_fluid, sim.n_wall) == (o.N, o.Nb)
    if squeezed:
        squeeze(sim, o)
    active = False
    for s in range(steps):
        sim.step_pbf(1)
        o.step_pbf(1)
        if s % 25 == 0 or s == steps - 1:
            lam = o.get(orc.F_PBF_LAMBDA)
            same(sim.download(nat.F_RHO), o.get(orc.F_RHO), "rho, step %d" % s)
            same(sim.download(nat.F_PBF_LAMBDA), lam, "pbf_lambda, step %d" % s)
            same(sim.download(nat.F_PBF_DELTA_POS), o.get(orc.F_PBF_DELTA_POS), "delta_pos, step %d" % s)
            same(sim.download(nat.F_POS), o.get(orc.F_POS), "pos, step %d" % s)
            same(sim.download(nat.F_VEL), o.get(orc.F_VEL), "vel, step %d" % s)
            active = active or bool((lam != 0).any())
    pos = sim.download(nat.F_POS)
    assert np.isfinite(pos).all()
    if squeezed or steps >= 400:
        assert active, "the density constraint never became active: lambda / delta_pos not exercised"
    sim.close(); o.close()


def test_pbf_mirror_api_and_30k():
    """main.py:65-68 discovery for `pbf`, and the 30 k dam break: the first steps against the oracle, then a longer run stays finite."""
    import importlib
    from cfd_taichi_amd import ParticleSystem
    cfg = scenes.get("breaking_dam_30k_pbf")
    ps = ParticleSystem(cfg)
    solver = getattr(importlib.import_module("cfd_taichi_amd.pbf_solver"), "pbf_solver")(ps, cfg)
    o = orc.Oracle(cfg, num_threads=8)
    for _ in range(5):
        solver.step(); o.step_pbf(1)
    same(ps.fluid_particles.pos.to_numpy(), o.get(orc.F_POS), "pos")
    same(ps.fluid_particles.vel.to_numpy(), o.get(orc.F_VEL), "vel")
    same(solver.pbf_lambda.to_numpy(), o.get(orc.F_PBF_LAMBDA), "lambda")
    same(solver.pos_predict.to_numpy(), o.get(orc.F_POS), "pos_predict == pos after a step (pbf_solver.py:84)")
    solver.step(400)
    pos = ps.fluid_particles.pos.to_numpy()
    assert np.isfinite(pos).all() and solver.delta_time[None] == np.float32(2.5e-4) and solver.simulate_cnt[None] == 405
    o.close()


def test_pbf_rejects_a_rigid_body():
    from cfd_taichi_amd import mesh
    cfg = scenes.get("dfsph_rigid_small")
    cfg["solver"]["name"] = "pbf"
    with pytest.raises(nat.SphError):
        nat.Simulation(nat.config_from_dict(cfg), rigid=mesh.rigid_from_config(cfg))


@pytest.mark.parametrize("scene", ["pbf_tiny_wall", "pbf_tiny_clamp"])
def test_pbf_compute_density_is_poly6_and_leaves_lambda_alone(scene):
    """solver_base.compute_all_rho() on a pbf solver: pbf_solver.py:166-174 overrides the rho callbacks with the poly6 kernel, and
    nothing but rho[] is written -- pbf_lambda keeps the values of the last step, positions and velocities keep their roles."""
    cfg = scenes.get(scene)
    sim = nat.Simulation(nat.config_from_dict(cfg))
    o = orc.Oracle(cfg, num_threads=8)
    squeeze(sim, o)
    for _ in range(3):
        sim.step_pbf(1); o.step_pbf(1)
    lam = o.get(orc.F_PBF_LAMBDA).copy()
    assert (lam != 0).any()
    pos, vel = sim.download(nat.F_POS), sim.download(nat.F_VEL)
    sim.compute_density()
    o.build_grid(); o.compute_rho()
    same(sim.download(nat.F_RHO), o.get(orc.F_RHO), "rho after compute_density")
    same(sim.download(nat.F_PBF_LAMBDA), lam, "pbf_lambda after compute_density")
    same(sim.download(nat.F_POS), pos, "pos after compute_density")
    same(sim.download(nat.F_VEL), vel, "vel after compute_density")
    # rho is the poly6 sum, not the cubic spline one: the two differ on a squeezed lattice
    sim.step_pbf(1); o.step_pbf(1)
    same(sim.download(nat.F_POS), o.get(orc.F_POS), "pos, the step after")
    same(sim.download(nat.F_PBF_LAMBDA), o.get(orc.F_PBF_LAMBDA), "pbf_lambda, the step after")
    sim.close(); o.close()
