"""GPU suite: the storage order of the cells is invisible in the results.

The library keeps the cells of large scenes along a Morton curve (tiles of 4 cells per axis, cell_slot() in
csrc/sph_kernels.h) and those of small scenes in the reference's 1-D order; every consumer still walks the 27 cells in the
reference's sequence, so both orders must give the same bits.  The other GPU suites run their small scenes in the linear
order; here the same kinds of scene (walls, clamp walls, particles that leak out of the box, a rigid body, slabs) are forced
onto the Morton curve and compared with the linear order and with the oracle.

On the curve the DFSPH sweeps also stage their gather operands in LDS (the list build writes, per workgroup, the set of particles it
can see and the neighbour lists in indices local to that set; workgroups whose set exceeds the capacity keep global indices).  That is
on by default, so every Morton case here runs it; the staging tests below compare it with SPH_STAGE=0 and with capacities small
enough that staged and unstaged workgroups mix."""
import numpy as np
import pytest

from cfd_taichi_amd import _native as nat
from cfd_taichi_amd import mesh, scenes
from oracle import oracle as orc
from test_fuzz_gpu import random_scene
from test_slab_gpu import run_slabs

pytestmark = pytest.mark.gpu

FIELDS = (nat.F_POS, nat.F_VEL, nat.F_RHO)


def make(cfg, order, monkeypatch, tile="4", rigid=None):
    monkeypatch.setenv("SPH_CELL_ORDER", order)
    monkeypatch.setenv("SPH_CELL_TILE", tile)
    return nat.Simulation(nat.config_from_dict(cfg), rigid=rigid)


@pytest.mark.parametrize("scene,steps,tile", [("wcsph_small", 80, "4"), ("dfsph_small", 40, "4"), ("dfsph_tiny_wall", 60, "16"), ("wcsph_tiny_wall", 120, "16"),
                                              ("dfsph_tiny_wall_pcisph", 40, "4"), ("dfsph_tiny_wall_iisph", 800, "4"), ("dfsph_dam_x", 1500, "4"),
                                              ("breaking_dam_30k_dfsph", 10, "4"), ("breaking_dam_30k_dfsph", 10, "16")])
def test_morton_and_linear_orders_agree(scene, steps, tile, monkeypatch):
    cfg = scenes.get(scene)
    a, b = make(cfg, "morton", monkeypatch, tile), make(cfg, "linear", monkeypatch)
    lost = 0
    for s in range(steps):
        sa, sb = a.step(1), b.step(1)
        if sa is not None:
            assert (sa.n_div, sa.n_dens, sa.div_err, sa.dens_err, sa.dt, sa.lost, sa.max_nbrs) == (
                sb.n_div, sb.n_dens, sb.div_err, sb.dens_err, sb.dt, sb.lost, sb.max_nbrs), (scene, s)
            lost = max(lost, sa.lost)
    for f in FIELDS:
        assert np.array_equal(a.download(f), b.download(f), equal_nan=True), (scene, f)
    if steps >= 800:             # the long cases are here for particles that leak through the walls: wrapped cell indices, the "outside" bucket
        # (particles that share a wrapped or "outside" cell index but not their cell coordinates once hung the staging plan's look-up)
        pos = a.download(nat.F_POS)
        outside = int(((pos < 0) | (pos > np.asarray(cfg["scene"]["box_max"], dtype=np.float32))).any(axis=1).sum())
        assert outside > 0 or lost > 0, scene
    a.close(); b.close()


def test_morton_device_order_small_scene(monkeypatch):
    """The sorted arrays follow the Morton curve of the cell coordinates, ascending id inside a cell."""
    cfg = scenes.get("dfsph_small")
    sim = make(cfg, "morton", monkeypatch)
    sim.step_dfsph(20)
    sim.build_neighbors()
    ids, lpos = sim.download_local(nat.F_POS)
    c3 = np.floor(lpos / np.float32(4 * cfg["scene"]["particle_radius"])).astype(np.int64)
    code = np.zeros(len(c3), dtype=np.int64)
    for k in range(10):
        for a in range(3):
            code |= ((c3[:, a] >> k) & 1) << (3 * k + a)
    assert np.all(np.diff(code) >= 0) and len(np.unique(code)) > 100
    same = np.diff(code) == 0
    assert np.all(np.diff(ids.astype(np.int64))[same] > 0)
    sim.close()


@pytest.mark.parametrize("solver", ["wcsph", "dfsph", "pcisph", "iisph"])
@pytest.mark.parametrize("seed", range(3))
def test_random_scene_on_the_morton_curve_matches_oracle(solver, seed, monkeypatch):
    rng = np.random.default_rng(3000 + seed)
    cfg = random_scene(rng, solver)
    sim = make(cfg, "morton", monkeypatch)
    o = orc.Oracle(cfg, solver=solver, num_threads=4)
    for s in range(25):
        sim.step(1)
        {"wcsph": o.step_wcsph, "pcisph": o.step_pcisph, "iisph": o.step_iisph}.get(solver, lambda n: o.step_dfsph(n, 100))(1)
    for f, of in ((nat.F_POS, orc.F_POS), (nat.F_VEL, orc.F_VEL), (nat.F_RHO, orc.F_RHO)):
        assert np.array_equal(sim.download(f), o.get(of), equal_nan=True), (solver, seed, f, cfg)
    sim.close(); o.close()


def test_rigid_coupling_on_the_morton_curve(monkeypatch):
    cfg = scenes.get("dfsph_rigid_small")
    rg = mesh.rigid_from_config(cfg)
    sim = make(cfg, "morton", monkeypatch, rigid=rg)
    o = orc.Oracle(cfg, num_threads=8, rigid=rg)
    for s in range(80):
        st = sim.step_dfsph(1)
        o.step_dfsph(1, 100)
        assert (st.n_div, st.n_dens, st.div_err, st.dt) == (o.last_stats.n_div, o.last_stats.n_dens, o.last_stats.div_err, o.last_stats.dt), s
        sim.rigid_step(); o.rigid_step()
    a, b = sim.rigid_scalars(), o.rigid_scalars()
    for k in ("centroid", "omega", "vel"):
        assert np.array_equal(np.float32(a[k]), np.float32(b[k])), k
    assert np.array_equal(sim.download(nat.F_POS), o.get(orc.F_POS)) and np.array_equal(sim.download(nat.F_VEL), o.get(orc.F_VEL))
    assert np.array_equal(sim.download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID), o.get(orc.F_RIGID_FORCE))
    sim.close(); o.close()


@pytest.mark.parametrize("scene,world,steps,rebalance", [("dfsph_dam_x", 3, 120, 7), ("dfsph_tiny_wall_iisph", 2, 200, 9)])
def test_slabs_on_the_morton_curve(tmp_path, monkeypatch, scene, world, steps, rebalance):
    """Edge and ghost lists enumerate cell columns in (y, z) order whatever the storage order: slabs still match one GPU."""
    monkeypatch.setenv("SPH_CELL_ORDER", "morton")
    r = run_slabs(tmp_path, scene, world, steps, rebalance=rebalance)
    assert r["pos_equal"] and r["vel_equal"] and r["rho_equal"] and r["stats_equal"], {k: r[k] for k in ("pos_rel_err", "slabs")}


@pytest.mark.parametrize("scene,steps", [("dfsph_small", 40), ("dfsph_tiny_wall", 60), ("dfsph_dam_x", 300), ("breaking_dam_30k_dfsph", 12)])
@pytest.mark.parametrize("cap", ["1664", "700", "300", "64"])
def test_lds_staging_is_invisible(scene, steps, cap, monkeypatch):
    """Same bits with the operands staged in LDS, at several capacities (64: almost every workgroup falls back to global indices)."""
    cfg = scenes.get(scene)
    monkeypatch.setenv("SPH_STAGE", "1")
    monkeypatch.setenv("SPH_STAGE_CAP", cap)
    a = make(cfg, "morton", monkeypatch)
    monkeypatch.setenv("SPH_STAGE", "0")
    b = make(cfg, "morton", monkeypatch)
    for s_ in range(steps):
        sa, sb = a.step(1), b.step(1)
        assert (sa.n_div, sa.n_dens, sa.div_err, sa.dens_err, sa.dt, sa.max_nbrs) == (sb.n_div, sb.n_dens, sb.div_err, sb.dens_err, sb.dt, sb.max_nbrs), (scene, s_)
    for f in FIELDS:
        assert np.array_equal(a.download(f), b.download(f), equal_nan=True), (scene, cap, f)
    a.close(); b.close()


@pytest.mark.parametrize("cap", ["1664", "300"])
def test_list_and_scalar_formats_are_invisible(cap, monkeypatch):
    """The storage formats of the staged DFSPH sweeps -- 16-bit local list indices (SPH_NL16) and k / rho in its own 4-byte array
    (SPH_KR_SPLIT) -- change no bit: all on, the scalar array off, both off, at a capacity where staged and unstaged workgroups mix."""
    cfg = scenes.get("breaking_dam_30k_dfsph")
    monkeypatch.setenv("SPH_STAGE_CAP", cap)
    sims = []
    for nl16, split in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("SPH_NL16", nl16)
        monkeypatch.setenv("SPH_KR_SPLIT", split)
        sims.append(make(cfg, "morton", monkeypatch))
    for s_ in range(15):
        st = [sim.step(1) for sim in sims]
        for other in st[1:]:
            assert (st[0].n_div, st[0].n_dens, st[0].div_err, st[0].dens_err, st[0].dt) == (other.n_div, other.n_dens, other.div_err, other.dens_err, other.dt), s_
    for f in FIELDS:
        ref = sims[0].download(f)
        assert all(np.array_equal(ref, sim.download(f)) for sim in sims[1:]), f
    for sim in sims:
        sim.close()


@pytest.mark.parametrize("scene,steps,cap", [("dfsph_rigid_small", 150, "1664"), ("dfsph_rigid_small", 80, "300"), ("dfsph_rigid_tilted", 60, "1664")])
def test_list_and_scalar_formats_with_a_rigid_body(scene, steps, cap, monkeypatch):
    """... and with a coupled body: the list build decides per workgroup -- 16-bit local indices where no cell of the workgroup's
    neighbourhood holds a rigid sample, 32-bit local indices with tagged rigid entries in the shell around the body, 32-bit global ones
    where the set did not fit (kStageLists16 in stage_cnt) -- and k / rho travels in its own array for all of them.  All on, the scalar
    array off, both off: coupled steps, force on the body and body state included."""
    cfg = scenes.get(scene)
    rg = mesh.rigid_from_config(cfg)
    monkeypatch.setenv("SPH_STAGE_CAP", cap)
    sims = []
    for nl16, split in (("1", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("SPH_NL16", nl16)
        monkeypatch.setenv("SPH_KR_SPLIT", split)
        sims.append(make(cfg, "morton", monkeypatch, rigid=rg))
    for s_ in range(steps):
        st = [sim.step_dfsph(1) for sim in sims]
        for other in st[1:]:
            assert (st[0].n_div, st[0].n_dens, st[0].div_err, st[0].dens_err, st[0].dt) == (other.n_div, other.n_dens, other.div_err, other.dens_err, other.dt), s_
        ref = sims[0].download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID)
        assert all(np.array_equal(ref, sim.download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID)) for sim in sims[1:]), s_
        for sim in sims:
            sim.rigid_step()
    for f in FIELDS + (nat.F_RHO_ADV, nat.F_WARM_K):
        ref = sims[0].download(f)
        assert all(np.array_equal(ref, sim.download(f)) for sim in sims[1:]), f
    ra = sims[0].rigid_scalars()
    for sim in sims[1:]:
        rb = sim.rigid_scalars()
        for k in ("centroid", "omega", "vel"):
            assert np.array_equal(np.float32(ra[k]), np.float32(rb[k])), k
    for sim in sims:
        sim.close()


@pytest.mark.parametrize("scene,steps,cap,arith", [("breaking_dam_30k_dfsph", 120, "1664", 0), ("dfsph_small", 150, "300", 0), ("dfsph_dam_x", 150, "1664", 0),
                                                   ("breaking_dam_30k_dfsph", 60, "1664", 1)])
def test_density_loop_change_propagation_is_invisible(scene, steps, cap, arith, monkeypatch):
    """Round 3: tiles of the constant-density loop whose inputs did not change are not recomputed (tile_nz / tile_dirty, sph_kernels.h:
    stage_sources_flagged).  With SPH_TILE_SKIP=0 every tile computes in every iteration: same state, iteration counts and residuals,
    in the exact arithmetic and in the relaxed one (whose skipped and computed tiles give the same bits for the same reason), also at a
    capacity where staged and unstaged workgroups mix.  The two must be compared with ==: a skipped correction leaves a velocity
    component of -0 where the computed one gives v - (-0) = +0.
    Round 6: the sweep that writes an operand tells the tiles that stage it to run (DensFlow) -- the default; SPH_DENS_PUSH=0 is round 3's
    form, in which every tile reads its staging plan to find out.  All three in lock step."""
    cfg = scenes.get(scene)
    monkeypatch.setenv("SPH_STAGE_CAP", cap)
    monkeypatch.setenv("SPH_CELL_ORDER", "morton")
    sims = []
    for skip, push, cap_rows in (("1", "1", None), ("1", "0", None), ("0", "1", None), ("1", "1", "6")):
        monkeypatch.setenv("SPH_TILE_SKIP", skip)
        monkeypatch.setenv("SPH_DENS_PUSH", push)
        if cap_rows:          # rows of at most six tiles: most tiles' rows say "unknown" -- they always run and, as producers, raise the broadcast word
            monkeypatch.setenv("SPH_NBR_CAP", cap_rows)
        sims.append(nat.Simulation(nat.config_from_dict(cfg, arith=arith)))
        monkeypatch.delenv("SPH_NBR_CAP", raising=False)
    n_dens = []
    for s_ in range(steps):
        a, b, c_, d_ = (sm.step_dfsph(1) for sm in sims)
        for o in (b, c_, d_):
            assert (a.n_div, a.n_dens, a.div_err, a.dens_err, a.dt) == (o.n_div, o.n_dens, o.div_err, o.dens_err, o.dt), (scene, s_)
        n_dens.append(a.n_dens)
    for f in FIELDS + (nat.F_RHO_ADV, nat.F_WARM_K, nat.F_ALPHA):
        for o in sims[1:]:
            assert np.array_equal(sims[0].download(f), o.download(f)), (scene, f)
    assert max(n_dens) >= 3, "the density loop never iterated past its minimum: nothing was exercised"
    for sim in sims:
        sim.close()


@pytest.mark.parametrize("scene,steps,order,cap,quad", [("breaking_dam_30k_dfsph", 80, "morton", "1664", "1"), ("dfsph_small", 100, "morton", "300", "1"),
                                                        ("dfsph_dam_x", 150, "morton", "1664", "1"), ("breaking_dam_30k_dfsph", 40, "linear", "1664", "0"),
                                                        ("dfsph_rigid_small", 100, "morton", "1664", "1"), ("dfsph_rigid_small", 60, "linear", "1664", "0")])
def test_wall_gradient_cache_is_invisible(scene, steps, order, cap, quad, monkeypatch):
    """Round 3: D1 leaves (grad W_ib, V_b) of every wall-list entry in a per-step cache and the sweeps of the solver loops read it back
    instead of gathering the wall particle and re-deriving the gradient (for_wall_cache, sph_kernels.h).  SPH_WALL_CACHE=0 makes them walk
    the wall lists as before: same state, iteration counts and residuals -- staged sweeps, mixed staged / unstaged capacity, the plain
    one-lane-per-particle sweeps in the reference's cell order, and with a rigid body in the lists."""
    cfg = scenes.get(scene)
    rg = mesh.rigid_from_config(cfg) if "rigid" in scene else None
    monkeypatch.setenv("SPH_STAGE_CAP", cap)
    monkeypatch.setenv("SPH_CELL_ORDER", order)
    monkeypatch.setenv("SPH_QUAD", quad)
    sims = []
    for on in ("1", "0"):
        monkeypatch.setenv("SPH_WALL_CACHE", on)
        sims.append(nat.Simulation(nat.config_from_dict(cfg), rigid=rg))
    for s_ in range(steps):
        a, b = sims[0].step_dfsph(1), sims[1].step_dfsph(1)
        assert (a.n_div, a.n_dens, a.div_err, a.dens_err, a.dt) == (b.n_div, b.n_dens, b.div_err, b.dens_err, b.dt), (scene, s_)
        if rg is not None:
            assert np.array_equal(sims[0].download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID), sims[1].download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID)), s_
            for sim in sims:
                sim.rigid_step()
    for f in FIELDS + (nat.F_RHO_ADV, nat.F_WARM_K, nat.F_ALPHA):
        assert np.array_equal(sims[0].download(f), sims[1].download(f)), (scene, f)
    for sim in sims:
        sim.close()


@pytest.mark.parametrize("scene,steps,cap", [("breaking_dam_30k_pcisph", 40, "1664"), ("dfsph_tiny_wall_pcisph", 150, "1664"), ("breaking_dam_30k_pcisph", 45, "300")])
def test_pcisph_change_propagation_is_invisible(scene, steps, cap, monkeypatch):
    """The same idea in the PCISPH pressure loop (sph_pressure_kernels.h, k_pci_press): tiles whose staged pressures are all 0 and whose
    outputs already hold the zero-pressure values skip update_press_force.  SPH_TILE_SKIP=0 computes everything: same state, pressures,
    iteration counts and residuals."""
    cfg = scenes.get(scene)
    monkeypatch.setenv("SPH_STAGE_CAP", cap)
    monkeypatch.setenv("SPH_CELL_ORDER", "morton")
    sims = []
    for skip in ("1", "0"):
        monkeypatch.setenv("SPH_TILE_SKIP", skip)
        sims.append(nat.Simulation(nat.config_from_dict(cfg)))
    iters = []
    for s_ in range(steps):
        a, b = sims[0].step_pcisph(1), sims[1].step_pcisph(1)
        assert (a.n_dens, a.dens_err) == (b.n_dens, b.dens_err), (scene, s_, a.n_dens, b.n_dens)
        iters.append(a.n_dens)
    for f in FIELDS + (nat.F_PRESS_ITER, nat.F_PRESS_FORCE, nat.F_POS_PREDICT):
        assert np.array_equal(sims[0].download(f), sims[1].download(f)), (scene, f)
    assert max(iters) >= 3
    for sim in sims:
        sim.close()


@pytest.mark.parametrize("scene,steps", [("iisph_config_backup", 150), ("dfsph_tiny_wall_iisph", 150)])
def test_iisph_tiles_without_pressure_are_invisible(scene, steps, monkeypatch):
    """... and in IISPH's compute_all_d_ij (k_ii_dij): a tile whose staged pressures are all 0 and whose d_ij already hold the zeros returns."""
    cfg = scenes.get(scene)
    monkeypatch.setenv("SPH_CELL_ORDER", "morton")
    sims = []
    for skip in ("1", "0"):
        monkeypatch.setenv("SPH_TILE_SKIP", skip)
        sims.append(nat.Simulation(nat.config_from_dict(cfg)))
    iters = []
    for s_ in range(steps):
        a, b = sims[0].step_iisph(1), sims[1].step_iisph(1)
        assert (a.n_dens, a.dens_err) == (b.n_dens, b.dens_err), (scene, s_, a.n_dens, b.n_dens)
        iters.append(a.n_dens)
    for f in FIELDS + (nat.F_PRESS_ITER, nat.F_D_IJ, nat.F_PRESS_FORCE):
        assert np.array_equal(sims[0].download(f), sims[1].download(f)), (scene, f)
    assert max(iters) >= 3
    for sim in sims:
        sim.close()


def test_density_loop_change_propagation_with_a_rigid_body(monkeypatch):
    """... and with rigid entries in the lists (the body's term of the correction is proportional to the particle's own stiffness, and the
    body is at rest within a solver loop): coupled steps with and without SPH_TILE_SKIP, body included."""
    cfg = scenes.get("dfsph_rigid_small")
    rg = mesh.rigid_from_config(cfg)
    monkeypatch.setenv("SPH_CELL_ORDER", "morton")
    sims = []
    for skip in ("1", "0"):
        monkeypatch.setenv("SPH_TILE_SKIP", skip)
        sims.append(nat.Simulation(nat.config_from_dict(cfg), rigid=rg))
    n_dens = []
    for s_ in range(120):
        a, b = sims[0].step_dfsph(1), sims[1].step_dfsph(1)
        assert (a.n_div, a.n_dens, a.div_err, a.dens_err, a.dt) == (b.n_div, b.n_dens, b.div_err, b.dens_err, b.dt), s_
        n_dens.append(a.n_dens)
        sims[0].rigid_step(); sims[1].rigid_step()
    for f in FIELDS + (nat.F_RHO_ADV, nat.F_WARM_K):
        assert np.array_equal(sims[0].download(f), sims[1].download(f)), f
    assert np.array_equal(sims[0].download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID), sims[1].download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID))
    ra, rb = sims[0].rigid_scalars(), sims[1].rigid_scalars()
    for k in ("centroid", "omega", "vel"):
        assert np.array_equal(np.float32(ra[k]), np.float32(rb[k])), k
    assert max(n_dens) >= 3
    for sim in sims:
        sim.close()


@pytest.mark.parametrize("cap", ["1664", "200"])
def test_lds_staging_with_a_rigid_body(cap, monkeypatch):
    """Tagged rigid entries stay global inside staged lists; the coupled run equals the unstaged one, body included."""
    cfg = scenes.get("dfsph_rigid_small")
    rg = mesh.rigid_from_config(cfg)
    monkeypatch.setenv("SPH_STAGE", "1")
    monkeypatch.setenv("SPH_STAGE_CAP", cap)
    a = make(cfg, "morton", monkeypatch, rigid=rg)
    monkeypatch.setenv("SPH_STAGE", "0")
    b = make(cfg, "morton", monkeypatch, rigid=rg)
    for s_ in range(100):
        sa, sb = a.step_dfsph(1), b.step_dfsph(1)
        assert (sa.n_div, sa.n_dens, sa.div_err, sa.dens_err, sa.dt) == (sb.n_div, sb.n_dens, sb.div_err, sb.dens_err, sb.dt), s_
        a.rigid_step(); b.rigid_step()
    for f in FIELDS:
        assert np.array_equal(a.download(f), b.download(f), equal_nan=True), f
    ra, rb = a.rigid_scalars(), b.rigid_scalars()
    for k in ("centroid", "omega", "vel"):
        assert np.array_equal(np.float32(ra[k]), np.float32(rb[k])), k
    a.close(); b.close()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("solver", ["dfsph", "iisph"])
def test_lds_staging_with_particles_outside_the_box(solver, monkeypatch):
    """Particles outside the grid share the "outside" bucket (or a wrapped cell index) while their cell coordinates -- and so the cells
    they walk -- differ: every run of equal COORDINATES must contribute its neighbourhood to the staging plan (a missing cell once
    hung the plan's look-up).  A few particles are put just outside each face of the box; staged and unstaged runs must agree."""
    cfg = scenes.get("dfsph_tiny_clamp")
    cfg["solver"]["name"] = solver
    monkeypatch.setenv("SPH_STAGE", "1")
    a = make(cfg, "morton", monkeypatch)
    monkeypatch.setenv("SPH_STAGE", "0")
    b = make(cfg, "morton", monkeypatch)
    pos = a.download(nat.F_POS)
    box = np.asarray(cfg["scene"]["box_max"], dtype=np.float32)
    h = np.float32(4 * cfg["scene"]["particle_radius"])
    rng = np.random.default_rng(3)
    pick = rng.choice(len(pos), size=48, replace=False)
    for n, i in enumerate(pick):
        axis, side = n % 3, (n // 3) % 2
        pos[i, axis] = (-np.float32(0.3) * h * (1 + n % 4)) if side == 0 else box[axis] + np.float32(0.3) * h * (1 + n % 4)
    for sim in (a, b):
        sim.upload(nat.F_POS, pos)
    lost = 0
    for s_ in range(12):
        sa, sb = a.step(1), b.step(1)
        assert (sa.n_div, sa.n_dens, sa.dens_err, sa.lost, sa.max_nbrs) == (sb.n_div, sb.n_dens, sb.dens_err, sb.lost, sb.max_nbrs), s_
        lost = max(lost, sa.lost)
    for f in FIELDS:
        assert np.array_equal(a.download(f), b.download(f), equal_nan=True), f
    assert lost > 0
    a.close(); b.close()


@pytest.mark.parametrize("seed", [1])
def test_random_scenes_on_slabs_on_the_morton_curve(tmp_path, monkeypatch, seed):
    """The seeded random scenes of test_slab_gpu (radius, box, water block, dt, wall model, solver) on 2-4 slabs, with the cells on the
    Morton curve and the DFSPH / IISPH sweeps staged, re-balanced every 3 steps."""
    import json as _json
    monkeypatch.setenv("SPH_CELL_ORDER", "morton")
    rng = np.random.default_rng(2100 + seed)
    solver = ["dfsph", "iisph", "dfsph"][seed]
    cfg = random_scene(rng, solver)
    cfg["scene"]["box_max"][0] = float(np.round(cfg["scene"]["box_max"][0] + 8 * 4 * cfg["scene"]["particle_radius"], 3))   # room for 4 slabs
    path = tmp_path / "scene.json"
    path.write_text(_json.dumps(cfg))
    world = int(rng.integers(2, 5))
    r = run_slabs(tmp_path, str(path), world, 60, rebalance=3)
    assert r["pos_equal"] and r["vel_equal"] and r["rho_equal"] and r["stats_equal"], (cfg, {k: r[k] for k in ("pos_rel_err", "slabs")})


@pytest.mark.parametrize("solver,dt,steps", [("wcsph", 2.5e-4, 120), ("pcisph", 2.5e-4, 80), ("iisph", 5e-4, 80)])
def test_rigid_coupling_of_the_other_solvers_on_the_morton_curve(solver, dt, steps, monkeypatch):
    """Rigid-coupled WCSPH / PCISPH / IISPH with the cells on the Morton curve (PCISPH and IISPH with staged sweeps, tagged rigid entries
    inside staged lists, the body's own neighbour lists) against the linear order: fluid and body bit for bit.  The lattice is squeezed
    towards the body like in test_rigid_gpu.make_solver, so that the pressure coupling acts from the first step."""
    cfg = scenes.get("dfsph_rigid_small")
    cfg["solver"]["name"] = solver
    cfg["solver"]["delta_time"] = dt
    rg = mesh.rigid_from_config(cfg)
    a = make(cfg, "morton", monkeypatch, rigid=rg)
    b = make(cfg, "linear", monkeypatch, rigid=rg)
    pos = a.download(nat.F_POS)
    about = np.array([pos[:, 0].max(), pos[:, 1].min(), 0.5 * (pos[:, 2].min() + pos[:, 2].max())], dtype=np.float32)
    squeezed = (about + (pos - about) * np.float32(0.86)).astype(np.float32)
    a.upload(nat.F_POS, squeezed); b.upload(nat.F_POS, squeezed)
    pushed = False
    for s_ in range(steps):
        a.step(1); b.step(1)
        fa, fb = a.download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID), b.download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID)
        assert np.array_equal(fa, fb), (solver, s_)
        pushed = pushed or float(np.abs(fa).max()) > 0
        a.rigid_step(); b.rigid_step()
    for f in FIELDS:
        assert np.array_equal(a.download(f), b.download(f), equal_nan=True), (solver, f)
    ra, rb = a.rigid_scalars(), b.rigid_scalars()
    for k in ("centroid", "omega", "vel"):
        assert np.array_equal(np.float32(ra[k]), np.float32(rb[k])), (solver, k)
    assert pushed, "the fluid never pushed the body: coupling not exercised"
    a.close(); b.close()


@pytest.mark.parametrize("waves", ["0", "3", "9"])
@pytest.mark.parametrize("solver,seed", [("wcsph", 0), ("dfsph", 1), ("iisph", 2)])
def test_list_build_variants_match_oracle(waves, solver, seed, monkeypatch):
    """Small unstaged scenes build their lists with k_build_nl_split (one wave per dx-plane or per (dx, dy) column of the same 64
    particles, two passes) and larger ones with k_build_nl; SPH_BNL_SPLIT forces one: each against the oracle on random scenes (the
    other suites run their small scenes on the size-picked variant only), neighbour counts and step statistics included."""
    monkeypatch.setenv("SPH_BNL_SPLIT", waves)
    rng = np.random.default_rng(4100 + seed)
    cfg = random_scene(rng, solver)
    sim = nat.Simulation(nat.config_from_dict(cfg))
    o = orc.Oracle(cfg, solver=solver, num_threads=4)
    for s in range(25):
        st = sim.step(1)
        {"wcsph": o.step_wcsph, "iisph": o.step_iisph}.get(solver, lambda n: o.step_dfsph(n, 100))(1)
        if solver != "wcsph":
            assert st.n_dens == o.last_stats.n_dens, (s, waves)
    for f, of in ((nat.F_POS, orc.F_POS), (nat.F_VEL, orc.F_VEL), (nat.F_RHO, orc.F_RHO)):
        assert np.array_equal(sim.download(f), o.get(of), equal_nan=True), (solver, seed, waves, f, cfg)
    sim.close(); o.close()


@pytest.mark.parametrize("waves", ["0", "3", "9"])
def test_list_build_variants_with_a_rigid_body(waves, monkeypatch):
    """The same three builds with rigid entries in the lists (tagged entries after a cell's fluid entries, the neighbour count with its
    rigid-entry quirk): coupled steps of the tilted box against the size-picked build, bit for bit, force on the body included."""
    cfg = scenes.get("dfsph_rigid_tilted")
    rg = mesh.rigid_from_config(cfg)
    sims = []
    for w in (waves, None):
        if w is None:
            monkeypatch.delenv("SPH_BNL_SPLIT")
        else:
            monkeypatch.setenv("SPH_BNL_SPLIT", w)
        sims.append(nat.Simulation(nat.config_from_dict(cfg), rigid=rg))
    a, b = sims
    for s in range(40):
        sa, sb = a.step(1), b.step(1)
        assert (sa.n_div, sa.n_dens, sa.dens_err, sa.max_nbrs) == (sb.n_div, sb.n_dens, sb.dens_err, sb.max_nbrs), s
        assert np.array_equal(a.download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID), b.download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID)), s
        a.rigid_step(); b.rigid_step()
    for f in FIELDS + (nat.F_NBR_COUNT,):
        assert np.array_equal(a.download(f), b.download(f), equal_nan=True), f
    a.close(); b.close()


@pytest.mark.parametrize("scene,steps", [("wcsph_small", 80), ("dfsph_small", 40), ("dfsph_tiny_clamp", 60), ("breaking_dam_30k_dfsph", 8),
                                         ("dfsph_tiny_wall_pcisph", 40), ("dfsph_tiny_wall_iisph", 200), ("pcisph_config_backup", 30)])
def test_quad_sweeps_are_invisible(scene, steps, monkeypatch):
    """Small unstaged scenes run their sweeps with four lanes per particle (lane q evaluates entry q of every group of four, the four
    terms are added in list order through DPP quad broadcasts; block partials per 64 particles, added in groups of four by the
    finalize): SPH_QUAD=0 keeps one lane per particle.  Same bits, same iteration counts and residuals, every step."""
    cfg = scenes.get(scene)
    monkeypatch.setenv("SPH_QUAD", "1")
    a = nat.Simulation(nat.config_from_dict(cfg))
    monkeypatch.setenv("SPH_QUAD", "0")
    b = nat.Simulation(nat.config_from_dict(cfg))
    for s_ in range(steps):
        sa, sb = a.step(1), b.step(1)
        if cfg["solver"]["name"] != "wcsph":
            assert (sa.n_div, sa.n_dens, sa.div_err, sa.dens_err, sa.dt, sa.max_nbrs) == (sb.n_div, sb.n_dens, sb.div_err, sb.dens_err, sb.dt, sb.max_nbrs), (scene, s_)
    for f in FIELDS:
        assert np.array_equal(a.download(f), b.download(f), equal_nan=True), (scene, f)
    a.close(); b.close()


def test_quad_sweeps_with_a_rigid_body(monkeypatch):
    cfg = scenes.get("dfsph_rigid_tilted")
    rg = mesh.rigid_from_config(cfg)
    sims = []
    for qd in ("1", "0"):
        monkeypatch.setenv("SPH_QUAD", qd)
        sims.append(nat.Simulation(nat.config_from_dict(cfg), rigid=rg))
    a, b = sims
    for s_ in range(40):
        sa, sb = a.step(1), b.step(1)
        assert (sa.n_div, sa.n_dens, sa.dens_err, sa.div_err) == (sb.n_div, sb.n_dens, sb.dens_err, sb.div_err), s_
        assert np.array_equal(a.download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID), b.download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID)), s_
        a.rigid_step(); b.rigid_step()
    for f in FIELDS:
        assert np.array_equal(a.download(f), b.download(f), equal_nan=True), f
    a.close(); b.close()


@pytest.mark.parametrize("scene,steps,order,cap,quad,arith", [
    ("dfsph_small", 150, "morton", "600", "1", 0), ("dfsph_tiny_clamp", 150, "linear", "1664", "0", 0), ("dfsph_dam_x", 150, "morton", "1664", "1", 0),
    ("breaking_dam_30k_dfsph", 15, "morton", "1664", "1", 0), ("dfsph_rigid_small", 60, "morton", "1664", "1", 0)])
def test_riding_loop_decisions_against_the_oracle(scene, steps, order, cap, quad, arith, monkeypatch):
    """One GPU: the loop decision after a residual sweep is taken by workgroup 0 of the correction launch behind it (fin_ride_block); in the
    divergence loop that correction runs AHEAD of the decision and is undone when the decision closes the loop (SpecSave / SpecUndo).  Every
    step whose divergence loop ends below its cap mispredicts once -- on these scenes most steps do.  Lock step with the oracle over thousands of
    decisions: iteration counts, residuals and dt in EVERY step (one stale partial or one missed undo would change a mean), the state bit for bit
    at the end; staged, mixed-capacity, plain and quad sweeps, tiles that skip (change propagation), with a rigid body."""
    cfg = scenes.get(scene)
    rg = mesh.rigid_from_config(cfg) if "rigid" in scene else None
    monkeypatch.setenv("SPH_STAGE_CAP", cap)
    monkeypatch.setenv("SPH_CELL_ORDER", order)
    monkeypatch.setenv("SPH_QUAD", quad)
    sim = nat.Simulation(nat.config_from_dict(cfg, arith=arith), rigid=rg)
    o = orc.Oracle(cfg, num_threads=8, rigid=rg)
    undone = 0
    for s_ in range(steps):
        a = sim.step_dfsph(1)
        o.step_dfsph(1, 100)
        b = o.last_stats
        assert (a.n_div, a.n_dens, a.n_div_evals, a.div_first_err, a.div_err, a.dens_err, a.dt) == \
               (b.n_div, b.n_dens, b.n_div_evals, b.div_first_err, b.div_err, b.dens_err, b.dt), (scene, s_)
        undone += 1 if a.n_div < 15 else 0
        if rg is not None and rg["active"]:
            sim.rigid_step(); o.rigid_step()
    assert undone > 0          # (the undo path ran)
    for f, fo in ((nat.F_POS, orc.F_POS), (nat.F_VEL, orc.F_VEL), (nat.F_RHO_ADV, orc.F_RHO_ADV), (nat.F_WARM_K, orc.F_WARM_K), (nat.F_ALPHA, orc.F_ALPHA)):
        assert np.array_equal(sim.download(f), o.get(fo)), (scene, f)
    sim.close(); o.close()
