"""The same scenario code, written once against the C-ABI of include/sph_mi355x.h, run on both implementations of its per-step path:
libsph_mi355x.so (HIP, the product) and oracle/liborc_abi.so (the CPU oracle behind the same entry points, SURVEY.md 8b "so tests can
swap backends").  The CPU part checks that the oracle's ABI library exports the path and agrees with the oracle's own interface."""
import numpy as np
import pytest

from cfd_taichi_amd import _native as nat
from cfd_taichi_amd import mesh, scenes
from oracle import oracle as orc


def scenario(lib, scene, steps):
    """A caller of the ABI: create, a few stages, steps with statistics, downloads.  Returns everything it saw."""
    cfg = scenes.get(scene)
    rigid = mesh.rigid_from_config(cfg) if cfg.get("solid") else None
    sim = nat.Simulation(nat.config_from_dict(cfg), rigid=rigid, lib=lib)
    seen = {"sizes": (sim.n_fluid, sim.n_wall, sim.n_rigid, tuple(sim.grid))}
    solver = cfg["solver"]["name"]
    if solver == "dfsph":
        sim.compute_alpha()
        seen["alpha0"] = sim.download(nat.F_ALPHA)
        seen["count0"] = sim.download(nat.F_NBR_COUNT)
    stats = []
    for _ in range(steps):
        st = sim.step(1)
        if st is not None:
            stats.append((st.n_div, st.n_dens, st.n_div_evals, st.capped, st.div_first_err, st.div_err, st.dens_err, st.dt))
        if rigid is not None and rigid["active"]:
            sim.rigid_step()
    seen["stats"] = stats
    for name, f in (("pos", nat.F_POS), ("vel", nat.F_VEL), ("rho", nat.F_RHO)):
        seen[name] = sim.download(f)
    if rigid is not None:
        seen["rigid_pos"] = sim.download(nat.F_RIGID_POS, nat.SPECIES_RIGID)
        seen["body"] = sim.rigid_scalars()
    seen["dt"] = sim.scalar(nat.S_DELTA_TIME)
    seen["cnt"] = sim.scalar(nat.S_SIMULATE_CNT)
    sim.close()
    return seen


def assert_same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], np.ndarray):
            assert np.array_equal(a[k], b[k]), k
        elif isinstance(a[k], dict):
            assert {n: np.float32(v).tolist() for n, v in a[k].items()} == {n: np.float32(v).tolist() for n, v in b[k].items()}, k
        else:
            assert a[k] == b[k], (k, a[k], b[k])


def test_oracle_abi_library_exports_the_path_and_matches_its_own_interface():
    lib = nat.bind_core(orc.ABI_LIB)                       # raises if a CORE_EXPORTS symbol is missing
    seen = scenario(lib, "dfsph_tiny_wall", 8)
    o = orc.Oracle(scenes.get("dfsph_tiny_wall"), num_threads=4)
    for _ in range(8):
        o.step_dfsph(1, 100)
    assert np.array_equal(seen["pos"], o.get(orc.F_POS)) and np.array_equal(seen["vel"], o.get(orc.F_VEL))
    assert seen["stats"][-1][:2] == (o.last_stats.n_div, o.last_stats.n_dens) and seen["cnt"] == 8
    o.close()


def test_solver_attributes_through_the_abi_on_the_oracle_backend():
    """sph_set_scalar(SPH_P_*) on the oracle's ABI library = Oracle.set_param on its own interface, and the attributes change what the loops do."""
    cfg = scenes.get("dfsph_tiny_wall")
    sim = nat.Simulation(nat.config_from_dict(cfg), lib=nat.bind_core(orc.ABI_LIB))
    o = orc.Oracle(cfg, num_threads=4)
    assert sim.param("max_iteration_density_divergence") == 15 and sim.param("tension_k") == 0.5 and sim.param("max_dt") == 1e-3
    for k, v in (("max_iteration_density_divergence", 2), ("density_divergence_threshold", 1.0), ("min_iteration_density", 5), ("tension_k", 1.5), ("max_dt", 5e-4)):
        sim.set_param(k, v); o.set_param(nat.SOLVER_PARAMS[k], v)
        assert sim.param(k) == float(v)
    for _ in range(12):
        st = sim.step(1); o.step_dfsph(1, 100)
        assert (st.n_div, st.n_dens, st.dt) == (o.last_stats.n_div, o.last_stats.n_dens, o.last_stats.dt)
        assert st.n_div <= 2 and st.n_dens >= 5 and st.dt <= np.float32(5e-4)
    assert np.array_equal(sim.download(nat.F_POS), o.get(orc.F_POS)) and np.array_equal(sim.download(nat.F_VEL), o.get(orc.F_VEL))
    sim.close(); o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene,steps", [("dfsph_small", 12), ("wcsph_tiny_wall", 40), ("dfsph_tiny_wall_pcisph", 15), ("dfsph_tiny_wall_iisph", 15),
                                         ("pbf_tiny_wall", 30), ("dfsph_rigid_small", 25)])
def test_same_scenario_on_both_backends(scene, steps):
    assert_same(scenario(None, scene, steps), scenario(nat.bind_core(orc.ABI_LIB), scene, steps))
