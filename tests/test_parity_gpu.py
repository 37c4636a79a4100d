"""GPU suite: the HIP path (through the C-ABI) against the CPU oracle on identical initial conditions.

Bar: north_star asks for positions/velocities within 1e-5 relative after N steps.  The HIP kernels
evaluate the same f32 expressions in the same order as the oracle (contraction off, IEEE divide/sqrt,
canonical neighbour order), so these tests assert the stronger property -- bit-for-bit equality --
and report the relative error that the 1e-5 bar applies to.
"""
import numpy as np
import pytest

from cfd_taichi_amd import _native as nat
from cfd_taichi_amd import scenes
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5   # north_star tolerance (relative, on positions / velocities)


def rel_err(a, b):
    scale = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) / scale


def make(scene, solver=None):
    cfg = scenes.get(scene)
    sim = nat.Simulation(nat.config_from_dict(cfg, solver_name=solver))
    o = orc.Oracle(cfg, solver=solver, num_threads=8)
    return sim, o


def assert_same(a, b, what):
    assert a.shape == b.shape, what
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError("%s differs at %d of %d entries; rel err %.3e; first %s: %r vs %r" % (
            what, len(bad), a.size, rel_err(a, b), bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))


def test_device_divide_sqrt_are_ieee():
    rng = np.random.default_rng(7)
    wide = (10.0 ** rng.uniform(-44, 38, 100000)).astype(np.float32)     # denormals to near-overflow
    edge = np.array([0.0, 1.0, 0.1, 1e-38, 1e-30, 1e30, 9.99e-31, 1.0001e30, 1e-45, 3.4e38], dtype=np.float32)
    a = np.concatenate([rng.uniform(1e-6, 10, 200000), rng.uniform(1e-30, 1e-20, 1000), wide, edge]).astype(np.float32)
    b = np.concatenate([rng.uniform(1e-3, 2000, 200000), rng.uniform(1e-3, 10, 1000), rng.uniform(1e-3, 10, len(wide)),
                        rng.uniform(1.0, 7.0, len(edge))]).astype(np.float32)
    with np.errstate(over="ignore", under="ignore"):
        quotient = (a / b).astype(np.float32)
    assert_same(nat.selftest_math(0, a, b), quotient, "a/b")
    assert_same(nat.selftest_math(1, a, b), np.sqrt(a).astype(np.float32), "sqrt(a)")
    # the sweeps' 11-instruction root (sph_device.h sqrt_rn): correctly rounded on [0, 2^63), denormals and perfect squares included
    s = np.concatenate([a[a < 9e18], (rng.integers(1, 4000, 50000).astype(np.float32) ** 2) * np.float32(2.0) ** rng.integers(-60, 20, 50000).astype(np.float32),
                        np.float32([0.0, 1e-45, 1.1754944e-38, 1.1754942e-38, 0.01, 0.010000001, 9.2e18])]).astype(np.float32)
    assert_same(nat.selftest_math(6, s, s), np.sqrt(s).astype(np.float32), "sqrt_rn(s)")


def test_wave_primitives():
    """The DPP / ds_swizzle / v_permlane32_swap butterflies of csrc/sph_device.h: lane 0 of a reduction holds exactly the tree a
    shfl_down ladder over 32, 16, ..., 1 builds (the f64 block partials depend on that association), the scan is the inclusive sum."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal(1024) * 10.0 ** rng.integers(-8, 9, 1024)
    tree = x.reshape(-1, 64).copy()
    for off in (32, 16, 8, 4, 2, 1):
        tree[:, :off] = tree[:, :off] + tree[:, off:2 * off]
    got = nat.selftest_wave(0, x).reshape(-1, 64)
    assert np.array_equal(got[:, 0], tree[:, 0])
    assert np.allclose(got, x.reshape(-1, 64).sum(1, keepdims=True), rtol=1e-9, atol=1e-9 * np.abs(x).max())   # every lane holds a full sum
    n = rng.integers(-1000, 1000, 1024).astype(np.float64)
    assert np.array_equal(nat.selftest_wave(1, n).reshape(-1, 64), np.repeat(n.reshape(-1, 64).sum(1, keepdims=True), 64, 1))
    f = x.astype(np.float32).astype(np.float64)
    assert np.array_equal(nat.selftest_wave(2, f).reshape(-1, 64), np.repeat(f.reshape(-1, 64).max(1, keepdims=True), 64, 1))
    assert np.array_equal(nat.selftest_wave(3, n).reshape(-1, 64), np.repeat(n.reshape(-1, 64).max(1, keepdims=True), 64, 1))
    assert np.array_equal(nat.selftest_wave(4, n).reshape(-1, 64), np.cumsum(n.reshape(-1, 64), axis=1))


def test_device_kernel_functions_match_oracle():
    rng = np.random.default_rng(11)
    # the device divides by h and by h*r with shortened exact sequences (sph_device.h): cover the whole support densely
    r = np.concatenate([rng.uniform(0, 0.12, 200000), 10.0 ** rng.uniform(-9, -1, 20000), [0.0, 0.05, 0.1, 0.1000001, 1e-7, 1e-12]]).astype(np.float32)
    h = np.full_like(r, 0.1)
    w = nat.selftest_math(2, r, h)
    w_ref = np.array([orc.cubic_kernel(float(x), 0.1) for x in r], dtype=np.float32)
    assert_same(w, w_ref, "cubic_kernel")
    x = np.concatenate([rng.uniform(-0.07, 0.07, 60000), 10.0 ** rng.uniform(-8, -2, 5000), [0.0, 0.05, 1e-7]]).astype(np.float32)
    y = np.concatenate([rng.uniform(-0.07, 0.07, 60000), -(10.0 ** rng.uniform(-8, -2, 5000)), [0.0, 0.0, 0.0]]).astype(np.float32)
    g = [nat.selftest_math(3 + k, x, y) for k in range(3)]
    ref = np.array([orc.cubic_kernel_derivative([a, b, np.float32(0.25) * a], 0.1) for a, b in zip(x, y)], dtype=np.float32)
    for k in range(3):
        assert_same(g[k], ref[:, k], "cubic_kernel_derivative[%d]" % k)


@pytest.mark.parametrize("scene", ["wcsph_tiny_wall", "dfsph_small", "breaking_dam_30k_wcsph"])
def test_initial_conditions(scene):
    sim, o = make(scene)
    assert (sim.n_fluid, sim.n_wall, tuple(sim.grid), sim.n_cells) == (o.N, o.Nb, tuple(o.grid), o.C)
    assert_same(sim.download(nat.F_POS), o.get(orc.F_POS), "fluid lattice")
    assert_same(sim.download(nat.F_WALL_POS, nat.SPECIES_WALL), o.get(orc.F_WALL_POS), "wall positions")
    assert_same(sim.download(nat.F_WALL_VOL, nat.SPECIES_WALL), o.get(orc.F_WALL_VOL), "wall volumes")
    sim.close(); o.close()


@pytest.mark.parametrize("scene", ["dfsph_tiny_wall", "dfsph_small", "breaking_dam_30k_dfsph"])
def test_density_alpha_neighbour_count_at_rest(scene, kats):
    sim, o = make(scene)
    sim.compute_alpha()
    o.compute_rho(); o.compute_alpha(); o.compute_nbr_count()
    assert_same(sim.download(nat.F_NBR_COUNT), o.get(orc.F_NBR_COUNT), "neighbour count")
    assert_same(sim.download(nat.F_RHO), o.get(orc.F_RHO), "rho")
    assert_same(sim.download(nat.F_ALPHA), o.get(orc.F_ALPHA), "alpha")
    sim.close(); o.close()


def test_density_on_perturbed_positions():
    # ragged cells: random displacements change cell occupancy and neighbour sets
    sim, o = make("dfsph_small")
    rng = np.random.default_rng(3)
    pos = o.get(orc.F_POS)
    pos = (pos + rng.uniform(-0.02, 0.02, pos.shape)).astype(np.float32)
    sim.upload(nat.F_POS, pos); o.set(orc.F_POS, pos)
    sim.compute_alpha()
    o.build_grid(); o.compute_rho(); o.compute_alpha(); o.compute_nbr_count()
    assert_same(sim.download(nat.F_POS), pos, "upload/download round trip")
    assert_same(sim.download(nat.F_NBR_COUNT), o.get(orc.F_NBR_COUNT), "neighbour count")
    assert_same(sim.download(nat.F_RHO), o.get(orc.F_RHO), "rho")
    assert_same(sim.download(nat.F_ALPHA), o.get(orc.F_ALPHA), "alpha")
    sim.close(); o.close()


@pytest.mark.parametrize("scene,steps", [("wcsph_tiny_wall", 200), ("wcsph_tiny_clamp", 200), ("wcsph_small", 100),
                                         ("breaking_dam_30k_wcsph", 60)])
def test_wcsph_steps(scene, steps):
    sim, o = make(scene)
    done = 0
    for chunk in (1, 4, steps - 5):
        sim.step_wcsph(chunk); o.step_wcsph(chunk); done += chunk
        p, po = sim.download(nat.F_POS), o.get(orc.F_POS)
        v, vo = sim.download(nat.F_VEL), o.get(orc.F_VEL)
        assert np.isfinite(p).all()
        assert rel_err(p, po) <= REL_TOL and rel_err(v, vo) <= REL_TOL, (done, rel_err(p, po), rel_err(v, vo))
        assert_same(p, po, "pos after %d steps" % done)
        assert_same(v, vo, "vel after %d steps" % done)
    assert_same(sim.download(nat.F_RHO), o.get(orc.F_RHO), "rho")
    assert_same(sim.download(nat.F_PRESSURE), o.get(orc.F_PRESSURE), "pressure")
    assert_same(sim.download(nat.F_ACC), o.get(orc.F_ACC), "acc")
    sim.close(); o.close()


@pytest.mark.parametrize("scene,steps", [("dfsph_tiny_wall", 60), ("dfsph_tiny_clamp", 60), ("dfsph_small", 40),
                                         ("breaking_dam_30k_dfsph", 12)])
def test_dfsph_steps(scene, steps):
    sim, o = make(scene)
    for s in range(steps):
        st = sim.step_dfsph(1)
        o.step_dfsph(1, 100)
        so = o.last_stats
        assert (st.n_div, st.n_dens, st.n_div_evals) == (so.n_div, so.n_dens, so.n_div_evals), (s, st.n_div, so.n_div, st.n_dens, so.n_dens)
        assert st.div_first_err == so.div_first_err and st.div_err == so.div_err and st.dens_err == so.dens_err, s
        assert st.dt == so.dt and st.lost == 0
    p, po = sim.download(nat.F_POS), o.get(orc.F_POS)
    v, vo = sim.download(nat.F_VEL), o.get(orc.F_VEL)
    assert rel_err(p, po) <= REL_TOL and rel_err(v, vo) <= REL_TOL
    assert_same(p, po, "pos"); assert_same(v, vo, "vel")
    for f_gpu, f_orc, name in ((nat.F_RHO, orc.F_RHO, "rho"), (nat.F_ALPHA, orc.F_ALPHA, "alpha"), (nat.F_WARM_K, orc.F_WARM_K, "warm_start_k"),
                               (nat.F_RHO_DER, orc.F_RHO_DER, "rho_derivative"), (nat.F_RHO_ADV, orc.F_RHO_ADV, "rho_adv"),
                               (nat.F_VEL_ADV, orc.F_VEL_ADV, "vel_adv")):
        assert_same(sim.download(f_gpu), o.get(f_orc), name)
    assert sim.scalar(nat.S_DELTA_TIME) == o.dt
    sim.close(); o.close()


def test_python_api_mirrors_reference_surface():
    from cfd_taichi_amd import ParticleSystem, dfsph_solver, wcsph_solver
    cfg = scenes.get("breaking_dam_30k_wcsph")
    cfg["solver"]["name"] = "iisph"          # what config/breaking_dam_30k.json really names; BASELINE overrides to wcsph
    ps = ParticleSystem(cfg)
    solver = wcsph_solver(ps, cfg)
    assert ps.particle_num == 29120 and ps.boundary_particles_num == 21602
    for _ in range(3):
        solver.step()
    pos = ps.fluid_particles.pos.to_numpy()
    assert pos.shape == (29120, 3) and pos.dtype == np.float32
    assert abs(solver.delta_time[None] - 2.5e-4) < 1e-10 and solver.simulate_cnt[None] == 3
    o = orc.Oracle(cfg, solver="wcsph", num_threads=8)
    o.step_wcsph(3)
    assert_same(pos, o.get(orc.F_POS), "pos via Python API")
    cfg2 = scenes.get("dfsph_small")
    ps2 = ParticleSystem(cfg2)
    s2 = dfsph_solver(ps2, cfg2)
    s2.step()
    assert abs(s2.delta_time[None] - 1e-3) < 1e-9 and abs(ps2.delta_time[None] - 1e-3) < 1e-9
    assert ps2.rgba.to_numpy().shape == (ps2.particle_num, 4)


def test_neighbour_overflow_is_reported():
    cfg = scenes.get("wcsph_tiny_wall")
    sim = nat.Simulation(nat.config_from_dict(cfg, max_neighbors=8))
    with pytest.raises(nat.SphError) as e:
        sim.step_wcsph(1)
    assert e.value.code == nat.SPH_E_OVERFLOW
    sim.close()


def test_headless_runner_writes_ply_and_obj(tmp_path):
    """The frame loop of main.py:165-205 without the window: the PLY payload is the device state (6 decimals, original particle order,
    the constant RGBA of ParticleSystem.py:152) and equals the oracle's positions after the same steps; the OBJ is `ps.mesh.export`
    of the vertices that followed the body."""
    import json
    from cfd_taichi_amd import mesh, run
    cfg = scenes.get("dfsph_rigid_small")
    cfg["scene"]["output_fps"] = 500
    path = tmp_path / "scene.json"
    path.write_text(json.dumps(cfg))
    frames, t, plys = run.main(["--config", str(path), "--steps", "6", "--ply-dir", str(tmp_path / "out")])
    assert frames == 6 and plys >= 2 and t > 0
    # the same frame loop on the oracle (main.py:165-173, 189): a frame is written whenever t / frame_time passes the frame counter
    rg = mesh.rigid_from_config(cfg)
    o = orc.Oracle(cfg, num_threads=4, rigid=rg)
    t_o, frame = 0.0, 0
    for _ in range(6):
        o.step_dfsph(1, 100); o.rigid_step()
        t_o += o.dt
        if not t_o / (1.0 / 500) > frame:
            continue
        text = (tmp_path / "out" / ("output_%06d.ply" % frame)).read_text().splitlines()
        assert text[0] == "ply" and text[1] == "format ascii 1.0" and text[3] == "element vertex 5759" and text[11] == "end_header"
        assert [l.split()[-1] for l in text[4:11]] == ["x", "y", "z", "red", "green", "blue", "alpha"]
        body = np.loadtxt(text[12:], dtype=np.float64)
        assert body.shape == (5759, 7)
        want = np.array([["%.6f" % v for v in row] for row in o.get(orc.F_POS)], dtype=np.float64)
        assert np.array_equal(body[:, :3], want), frame
        assert np.allclose(body[:, 3:], [0.0, 0.26, 0.68, 1.0], atol=1e-6)
        obj = (tmp_path / "out" / ("obj_%06d.obj" % frame)).read_text().splitlines()
        v = np.array([[float(x) for x in l.split()[1:]] for l in obj if l.startswith("v ")])
        assert v.shape == (8, 3) and sum(1 for l in obj if l.startswith("f ")) == 12
        assert np.allclose(v, o.get(orc.F_RIGID_VERT), atol=1e-7), frame
        frame += 1
    assert frame == plys and t_o == t
    o.close()


def test_mirror_api_prologue_and_mesh_export():
    """The pieces of the caller contract beside step(): solver_base.step() (the prologue alone, solver_base.py:136-143), the prints
    the reference makes every step (dfsph_solver.py:233, :416), ps.mesh.export (main.py:196-200)."""
    import contextlib
    import io
    from cfd_taichi_amd import ParticleSystem, dfsph_solver, rigid_solver, solver_base
    cfg = scenes.get("dfsph_rigid_small")
    ps = ParticleSystem(cfg)
    solver = dfsph_solver(ps, cfg)
    rs = rigid_solver(ps, cfg)
    o = orc.Oracle(cfg, num_threads=4, rigid=mesh_rigid(cfg))
    solver_base.step(solver)                               # count + grid rebuild + reset()
    assert solver.simulate_cnt[None] == 1
    o.compute_nbr_count()
    assert np.array_equal(ps.get_neighbour_count(), o.get(orc.F_NBR_COUNT).astype(np.int32))
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        solver.step(); rs.step()
    o.step_dfsph(1, 100); o.rigid_step()
    lines = out.getvalue().splitlines()
    assert lines[0] == "[divergence iteration] count: {}, first error {}, error {}".format(o.last_stats.n_div, solver.last_stats.div_first_err, solver.last_stats.div_err)
    assert lines[1].startswith("[density iteration] count: %d, error " % o.last_stats.n_dens)
    assert solver.simulate_cnt[None] == 2
    text = ps.mesh.export(file_type="obj")
    before = np.array([[float(x) for x in l.split()[1:]] for l in text.splitlines() if l.startswith("v ")])
    ps.update_mesh_vextics()
    after = np.array([[float(x) for x in l.split()[1:]] for l in ps.mesh.export(file_type="obj").splitlines() if l.startswith("v ")])
    assert np.allclose(after, o.get(orc.F_RIGID_VERT), atol=1e-7) and not np.allclose(before, after)     # placed and moved with the body
    assert text.count("\nf ") == 12
    o.close()


def mesh_rigid(cfg):
    from cfd_taichi_amd import mesh
    return mesh.rigid_from_config(cfg)


@pytest.mark.parametrize("solver", ["wcsph", "dfsph"])
def test_particles_outside_the_grid(solver):
    """Escaped particles (the reference only prints an error, ParticleSystem.py:393-395): both sides keep them out of the
    cell lists, still integrate them, and report how many there are."""
    scene = "wcsph_tiny_wall" if solver == "wcsph" else "dfsph_tiny_wall"
    sim, o = make(scene)
    pos = o.get(orc.F_POS)
    pos[0] = [-0.31, 0.2, 0.2]       # linear cell id < 0
    pos[1] = [0.5, 7.5, 0.5]         # far above the grid: id >= C
    pos[2] = [0.5, 0.5, -0.05]       # one negative component, id still valid (wraps into another cell): stays in the lists
    pos[3] = [1.15, 0.5, 0.5]        # beyond box_max.x but inside the +1 cell margin
    sim.upload(nat.F_POS, pos); o.set(orc.F_POS, pos)
    for _ in range(5):
        if solver == "wcsph":
            sim.step_wcsph(1); o.step_wcsph(1)
        else:
            st = sim.step_dfsph(1); o.step_dfsph(1, 100)
            assert st.lost == o.lost and st.lost >= 1
        assert_same(sim.download(nat.F_POS), o.get(orc.F_POS), "pos")
        assert_same(sim.download(nat.F_VEL), o.get(orc.F_VEL), "vel")
    assert_same(sim.download(nat.F_RHO), o.get(orc.F_RHO), "rho")
    sim.close(); o.close()


@pytest.mark.parametrize("scene,pre,post", [("dfsph_small", 40, 6), ("wcsph_small", 150, 20)])
def test_oracle_continues_from_device_state(scene, pre, post):
    """The hand-over bench.py's cpu_baseline uses: the device runs `pre` steps, its positions, velocities (dfsph: warm_start_k and
    delta_time too) are handed to a fresh oracle, and both continue -- bit for bit the same.  Everything else a step reads is
    recomputed from those (grid, rho, alpha), so the state is complete."""
    cfg = scenes.get(scene)
    sim = nat.Simulation(nat.config_from_dict(cfg))
    dfsph = scene.startswith("dfsph")
    if dfsph:
        sim.step_dfsph(pre)
    else:
        sim.step_wcsph(pre)
    o = orc.Oracle(cfg, num_threads=4)
    o.set(orc.F_POS, sim.download(nat.F_POS)); o.set(orc.F_VEL, sim.download(nat.F_VEL))
    if dfsph:
        o.set(orc.F_WARM_K, sim.download(nat.F_WARM_K)); o.set_dt(sim.scalar(nat.S_DELTA_TIME))
    for _ in range(post):
        if dfsph:
            st = sim.step_dfsph(1)
            o.step_dfsph(1, 100)
            assert (st.n_div, st.n_dens, st.div_err, st.dens_err, st.dt) == (o.last_stats.n_div, o.last_stats.n_dens, o.last_stats.div_err,
                                                                            o.last_stats.dens_err, o.last_stats.dt)
        else:
            sim.step_wcsph(1); o.step_wcsph(1)
    assert_same(sim.download(nat.F_POS), o.get(orc.F_POS), "pos")
    assert_same(sim.download(nat.F_VEL), o.get(orc.F_VEL), "vel")
    sim.close(); o.close()
