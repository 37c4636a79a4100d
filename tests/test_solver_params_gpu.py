"""GPU suite: `solver.<attribute> = value` reaches the kernels (VERDICT r4 missing #4).

The reference's thresholds are Python attributes of the solver object (dfsph_solver.py:21-29, solver_base.py:23-26, wcsph_solver.py:17-20): its
Python-scope loops read them at every step (:225, :396-404) and Taichi bakes the ones it meets inside a kernel (:113-117, solver_base.py:187-188, :216)
when that kernel first compiles.  A caller who edits them before the first step() gets the edited behaviour there; here they travel through
sph_set_scalar(SPH_P_*), and the library given the same values as the oracle stays bit-equal to it -- state, iteration counts, residuals, delta_time.
"""
import numpy as np
import pytest

from cfd_taichi_amd import _native as nat
from cfd_taichi_amd import scenes
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def same(a, b, what):
    assert a.shape == b.shape and np.array_equal(a, b), "%s differs at %d of %d entries" % (what, int((a != b).sum()), a.size)


def lockstep_dfsph(sim, o, steps):
    counts = []
    for s in range(steps):
        st = sim.step_dfsph(1)
        o.step_dfsph(1, 100)
        so = o.last_stats
        assert (st.n_div, st.n_dens, st.n_div_evals) == (so.n_div, so.n_dens, so.n_div_evals), (s, st.n_div, so.n_div, st.n_dens, so.n_dens)
        assert st.div_first_err == so.div_first_err and st.div_err == so.div_err and st.dens_err == so.dens_err and st.dt == so.dt, s
        counts.append((st.n_div, st.n_dens, st.dt))
    for f, fo, what in ((nat.F_POS, orc.F_POS, "pos"), (nat.F_VEL, orc.F_VEL, "vel"), (nat.F_WARM_K, orc.F_WARM_K, "warm_start_k"), (nat.F_RHO_ADV, orc.F_RHO_ADV, "rho_adv")):
        same(sim.download(f), o.get(fo), what)
    return counts


PARAM_SETS = {
    # the case VERDICT r4 names: a looser divergence threshold and a shorter cap, set before the first step
    "div_loop": {"density_divergence_threshold": 300.0, "max_iteration_density_divergence": 4},
    "div_min": {"min_iteration_density_divergence": 3, "density_divergence_threshold": 1e9},
    "dens_loop": {"density_threshold": 0.5, "min_iteration_density": 4},
    "no_warm_start": {"warm_start": 0},
    "fixed_dt": {"adaptive_dt": 0},
    "dt_window": {"max_dt": 4e-4, "min_dt": 2e-4},
    "fluid": {"viscosity_c_s": 20.0, "viscosity_alpha": 0.05, "viscosity_epsilon": 0.02, "tension_k": 1.5},
    "no_div_loop": {"max_iteration_density_divergence": 0},
    # the "no cap" idiom: the library enqueues such a loop in chunks of 32 iterations and looks at the loop state in between
    "no_cap": {"max_iteration_density_divergence": 1000},
}


CASES = [("dfsph_small", 25, None, name) for name in sorted(PARAM_SETS) if name != "no_cap"] \
    + [("dfsph_small", 25, "morton", name) for name in ("div_loop", "no_warm_start", "fluid", "dt_window")] \
    + [("dfsph_tiny_wall", 40, None, name) for name in ("dens_loop", "fixed_dt")] \
    + [("dfsph_small", 8, "morton", "no_cap")]          # (hundreds of divergence iterations per step: the staged path only)


@pytest.mark.parametrize("scene,steps,order,name", CASES)
def test_dfsph_attributes_set_before_the_first_step(scene, steps, order, name, monkeypatch):
    if order:
        monkeypatch.setenv("SPH_CELL_ORDER", order)          # the staged sweeps (Morton curve, LDS staging, 16-bit lists) of the large scenes
    cfg = scenes.get(scene)
    sim = nat.Simulation(nat.config_from_dict(cfg))
    o = orc.Oracle(cfg, num_threads=8)
    for k, v in PARAM_SETS[name].items():
        sim.set_param(k, v)
        o.set_param(nat.SOLVER_PARAMS[k], v)
        assert sim.param(k) == float(v)
    ref = nat.Simulation(nat.config_from_dict(cfg))
    if name == "fixed_dt":          # (the scene's delta_time is the adaptive rule's max_dt already: start from another one, which then must stay)
        sim.set_dt(5e-4); o.set_dt(5e-4); ref.set_dt(5e-4)
    counts = lockstep_dfsph(sim, o, steps)
    # ... and the edit did something: the same steps with the reference's values differ in the quantity the attribute governs
    ref_counts = [(st.n_div, st.n_dens, st.dt) for st in (ref.step_dfsph(1) for _ in range(steps))]
    assert counts != ref_counts or not np.array_equal(sim.download(nat.F_VEL), ref.download(nat.F_VEL)), name
    if name == "div_loop":
        assert max(c[0] for c in counts) <= 4
    if name == "no_div_loop":
        assert all(c[0] == 0 for c in counts)
    if name == "no_cap":
        assert max(c[0] for c in counts) > 32, "the divergence loop never crossed a chunk boundary: %r" % [c[0] for c in counts]
    if name == "fixed_dt":
        assert all(c[2] == np.float32(5e-4) for c in counts) and ref_counts[-1][2] != np.float32(5e-4)
    if name == "dt_window":
        assert all(np.float32(2e-4) <= c[2] <= np.float32(4e-4) for c in counts)
    sim.close(); o.close(); ref.close()


def test_live_attributes_can_change_between_steps():
    """The loop attributes are read by Python-scope loops (dfsph_solver.py:225, :400): an edit between two steps takes effect at the next one."""
    cfg = scenes.get("dfsph_small")
    sim = nat.Simulation(nat.config_from_dict(cfg))
    o = orc.Oracle(cfg, num_threads=8)
    lockstep_dfsph(sim, o, 8)
    for k, v in (("max_iteration_density_divergence", 3), ("density_threshold", 0.02)):
        sim.set_param(k, v); o.set_param(nat.SOLVER_PARAMS[k], v)
    counts = lockstep_dfsph(sim, o, 8)
    assert max(c[0] for c in counts) <= 3
    sim.close(); o.close()


@pytest.mark.parametrize("scene,steps", [("wcsph_small", 60), ("wcsph_tiny_wall", 100)])
def test_wcsph_viscosity_and_tension(scene, steps):
    cfg = scenes.get(scene)
    sim = nat.Simulation(nat.config_from_dict(cfg))
    o = orc.Oracle(cfg, num_threads=8)
    assert sim.param("viscosity_c_s") == 10.0 and sim.param("tension_k") == 0.2          # wcsph_solver.py:18, :20
    sim.step_wcsph(4); o.step_wcsph(4)                # (captured step pairs carry the constants: an edit must invalidate them)
    for k, v in (("viscosity_c_s", 30.0), ("tension_k", 2.0), ("viscosity_alpha", 0.2)):
        sim.set_param(k, v); o.set_param(nat.SOLVER_PARAMS[k], v)
    sim.step_wcsph(steps); o.step_wcsph(steps)
    same(sim.download(nat.F_POS), o.get(orc.F_POS), "pos")
    same(sim.download(nat.F_VEL), o.get(orc.F_VEL), "vel")
    ref = nat.Simulation(nat.config_from_dict(cfg))
    ref.step_wcsph(4 + steps)
    assert not np.array_equal(ref.download(nat.F_VEL), sim.download(nat.F_VEL))
    sim.close(); o.close(); ref.close()


@pytest.mark.parametrize("scene,solver,steps", [("dfsph_tiny_wall_pcisph", "pcisph", 25), ("dfsph_tiny_wall_iisph", "iisph", 25)])
def test_pressure_solvers_viscosity_and_tension(scene, solver, steps):
    """pcisph_solver / iisph_solver inherit viscosity_c_s = 13, tension_k = 0.5 from solver_base (:23-26); an edit before the first step reaches their
    external-force sweeps too: iteration counts, residuals and state bit-equal to the oracle under the same values."""
    cfg = scenes.get(scene)
    sim = nat.Simulation(nat.config_from_dict(cfg))
    o = orc.Oracle(cfg, solver=solver, num_threads=8)
    assert sim.param("viscosity_c_s") == 13.0 and sim.param("tension_k") == 0.5
    for k, v in (("viscosity_c_s", 25.0), ("tension_k", 1.25), ("viscosity_epsilon", 0.02)):
        sim.set_param(k, v); o.set_param(nat.SOLVER_PARAMS[k], v)
    for s_ in range(steps):
        st = sim.step(1)
        (o.step_pcisph if solver == "pcisph" else o.step_iisph)(1)
        assert (st.n_dens, st.dens_err) == (o.last_stats.n_dens, o.last_stats.dens_err), s_
    same(sim.download(nat.F_POS), o.get(orc.F_POS), "pos")
    same(sim.download(nat.F_VEL), o.get(orc.F_VEL), "vel")
    ref = nat.Simulation(nat.config_from_dict(cfg))
    for _ in range(steps):
        ref.step(1)
    assert not np.array_equal(ref.download(nat.F_VEL), sim.download(nat.F_VEL))
    with pytest.raises(nat.SphError):
        sim.set_param("max_dt", 5e-4)                        # a dfsph_solver attribute
    sim.close(); o.close(); ref.close()


def test_relaxed_arithmetic_takes_the_attributes_too(monkeypatch):
    """The tolerance-grade sweeps read the same attributes (warm_start in k_correct_rx, the loop parameters next to the loop state): under edited values
    the relaxed handle takes the same loop decisions as the exact one for the first steps from rest, where the two arithmetics agree to 1e-6."""
    monkeypatch.setenv("SPH_CELL_ORDER", "morton")
    cfg = scenes.get("dfsph_small")
    ex = nat.Simulation(nat.config_from_dict(cfg))
    rx = nat.Simulation(nat.config_from_dict(cfg, arith=nat.ARITH_RELAXED))
    for k, v in (("warm_start", 0), ("max_iteration_density_divergence", 3), ("min_iteration_density", 4), ("max_dt", 5e-4)):
        ex.set_param(k, v); rx.set_param(k, v)
    for s_ in range(6):
        a, b = ex.step_dfsph(1), rx.step_dfsph(1)
        assert (a.n_div, a.n_dens, a.dt) == (b.n_div, b.n_dens, b.dt) and a.n_div <= 3 and a.n_dens >= 4 and a.dt <= np.float32(5e-4), (s_, a.n_div, b.n_div, a.n_dens, b.n_dens)
    assert rx.scalar(nat.S_ARITH_RELAXED) == 1.0 and ex.scalar(nat.S_ARITH_RELAXED) == 0.0
    p, q = ex.download(nat.F_POS), rx.download(nat.F_POS)
    assert float(np.abs(p - q).max() / np.abs(p).max()) <= 1e-5
    ex.close(); rx.close()


def test_mirror_classes_forward_the_attributes():
    """The drop-in classes: `solver.density_divergence_threshold = ...` before the first step(), as a caller of the reference would write it."""
    from cfd_taichi_amd import ParticleSystem, dfsph_solver
    cfg = scenes.get("dfsph_small")
    cfg = dict(cfg, solver=dict(cfg["solver"], name="dfsph"))
    ps = ParticleSystem(cfg)
    solver = dfsph_solver(ps, cfg, verbose=False)
    solver.density_divergence_threshold = 300.0
    solver.max_iteration_density_divergence = 4
    solver.tension_k = 1.5
    o = orc.Oracle(cfg, num_threads=8)
    for k, v in (("density_divergence_threshold", 300.0), ("max_iteration_density_divergence", 4), ("tension_k", 1.5)):
        o.set_param(nat.SOLVER_PARAMS[k], v)
    for s in range(10):
        st = solver.step()
        o.step_dfsph(1, 100)
        assert (st.n_div, st.n_dens) == (o.last_stats.n_div, o.last_stats.n_dens) and st.n_div <= 4
        if s == 4:
            solver.tension_k = 9.0                      # baked into the reference's kernel at the first step: ignored from then on
            solver.max_iteration_density_divergence = 2 # read by a Python-scope loop: takes effect
            o.set_param(nat.SOLVER_PARAMS["max_iteration_density_divergence"], 2)
    assert st.n_div <= 2
    same(ps.fluid_particles.pos.to_numpy(), o.get(orc.F_POS), "pos")
    same(ps.fluid_particles.vel.to_numpy(), o.get(orc.F_VEL), "vel")
    o.close()


def test_bad_values_are_refused():
    sim = nat.Simulation(nat.config_from_dict(scenes.get("dfsph_small")))
    for k, v in (("max_dt", 0.0), ("min_dt", -1.0), ("max_iteration_density_divergence", 2.5), ("min_iteration_density", -1), ("viscosity_epsilon", float("inf")), ("tension_k", float("nan"))):
        before = sim.param(k)
        with pytest.raises(nat.SphError):
            sim.set_param(k, v)
        assert sim.param(k) == before, k             # a refused value leaves the attribute as it was
    # what the reference would run is taken: it assigns Python attributes without looking at them
    for k, v in (("viscosity_epsilon", 0.0), ("tension_k", -0.25), ("density_threshold", -1.0)):
        sim.set_param(k, v)
        assert sim.param(k) == v
    sim.close()
    b = nat.Simulation(nat.config_from_dict(scenes.get("pbf_tiny_wall")))
    before = b.param("tension_k")
    with pytest.raises(nat.SphError):
        b.set_param("tension_k", 0.3)                 # pbf_solver has no such attribute: refused BEFORE anything is written
    assert b.param("tension_k") == before
    b.close()
    w = nat.Simulation(nat.config_from_dict(scenes.get("wcsph_small")))
    with pytest.raises(nat.SphError):
        w.set_param("density_threshold", 0.2)             # a dfsph_solver attribute
    w.close()
