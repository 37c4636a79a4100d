"""GPU suite: seeded random scenes (particle radius, box, water block, start position, dt, wall model) x the four solvers, against the
oracle bit for bit.  Every BASELINE config uses r = 0.025; the constants derived from r (h, m, the exact division by h, the cell grid) and the
lattice arithmetic must hold for other radii and for boxes that are not multiples of the cell size."""
import numpy as np
import pytest

from cfd_taichi_amd import _native as nat
from cfd_taichi_amd import scenes
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def random_scene(rng, solver):
    r = float(rng.choice([0.02, 0.025, 0.03, 0.04, 0.05]))
    d, h = 2 * r, 4 * r
    box = [float(np.round(rng.uniform(10 * h, 18 * h), 3)) for _ in range(3)]
    walls = bool(rng.integers(0, 2))
    start = [float(np.round(rng.uniform(d, 3 * d), 4)) for _ in range(3)]
    water = [float(np.round(min(rng.uniform(5 * d, 11 * d), box[a] - start[a] - 2 * d), 4)) for a in range(3)]
    cfg = scenes.get("dfsph_tiny_wall")
    cfg["scene"].update(box_max=box, particle_radius=r)
    cfg["solver"].update(name=solver, boundary_handle=walls,
                         delta_time=float(rng.choice([2.5e-4, 5e-4])) if solver == "wcsph" else float(rng.choice([5e-4, 1e-3])))
    cfg["fluid"].update(start_pos=start, water_size=water)
    return cfg


@pytest.mark.parametrize("solver", ["wcsph", "dfsph", "pcisph", "iisph"])
@pytest.mark.parametrize("seed", range(6))
def test_random_scene_matches_oracle(solver, seed):
    rng = np.random.default_rng(1000 + seed)
    cfg = random_scene(rng, solver)
    sim = nat.Simulation(nat.config_from_dict(cfg))
    o = orc.Oracle(cfg, solver=solver, num_threads=4)
    assert (sim.n_fluid, sim.n_wall, tuple(sim.grid)) == (o.N, o.Nb, tuple(o.grid)), cfg
    assert 50 <= o.N <= 4000, o.N
    assert np.array_equal(sim.download(nat.F_POS), o.get(orc.F_POS)), "initial lattice"
    if o.Nb:
        assert np.array_equal(sim.download(nat.F_WALL_VOL, nat.SPECIES_WALL), o.get(orc.F_WALL_VOL)), "wall volumes"
    steps = 30
    for s in range(steps):
        if solver == "wcsph":
            sim.step_wcsph(1); o.step_wcsph(1)
        elif solver == "dfsph":
            st = sim.step_dfsph(1); o.step_dfsph(1, 100)
            so = o.last_stats
            assert (st.n_div, st.n_dens, st.div_err, st.dt) == (so.n_div, so.n_dens, so.div_err, so.dt), (s, cfg)
        else:
            st = sim.step(1)
            (o.step_pcisph if solver == "pcisph" else o.step_iisph)(1)
            assert (st.n_dens, st.dens_err) == (o.last_stats.n_dens, o.last_stats.dens_err), (s, cfg)
    for f, of in ((nat.F_POS, orc.F_POS), (nat.F_VEL, orc.F_VEL), (nat.F_RHO, orc.F_RHO)):
        a, b = sim.download(f), o.get(of)
        assert np.array_equal(a, b, equal_nan=True), (solver, seed, f, int((a != b).sum()), cfg)
    sim.close(); o.close()
