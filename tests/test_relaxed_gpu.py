"""GPU suite: SphConfig.arith = SPH_ARITH_RELAXED (csrc/sph_relaxed_kernels.h) -- the tolerance-grade dfsph sweeps.

The exact sweeps equal the oracle bit for bit.  The relaxed ones (approximate reciprocal square root, FMAs, gradient as one scalar
times the difference vector) cannot; what they have to meet is north_star's bar -- 1e-5 relative -- where that bar means something,
and the reference's OWN reproducibility where it does not: the reference's cell lists are filled by a racing parallel loop, every
neighbour sum has whatever order the thread schedule produced, and tools/envelope.py (profiles/r03/envelope_*.json) shows that two
legal executions of the reference itself are 1e-5 apart (max norm) after 5-10 steps and 1e-3 after 50: discrete gates -- list
membership at r = h, which the rest lattice hits exactly; the `neighbour count < 20` skip; max(., 0) -- turn one ulp into a
different sum.  So:
  * first steps: relaxed within 1e-5 (max norm) of the canonical oracle,
  * after 100 steps: relaxed deviates from the canonical oracle no more than seeded legal executions of the oracle do
    (per-particle quantiles), and
  * on dfsph_1m from the step-55 state: 20 steps next to the exact kernels -- iteration counts, quantiles, health.
The small scenes are put on the Morton curve (SPH_CELL_ORDER=morton: staged sweeps, 16-bit lists) -- the path relaxed covers."""
import os

import numpy as np
import pytest

from cfd_taichi_amd import _native as nat
from cfd_taichi_amd import scenes
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def make(scene, arith, morton=True):
    cfg = scenes.get(scene)
    old = os.environ.get("SPH_CELL_ORDER")
    if morton:
        os.environ["SPH_CELL_ORDER"] = "morton"
    try:
        sim = nat.Simulation(nat.config_from_dict(cfg, arith=arith))
    finally:
        if morton:
            if old is None:
                os.environ.pop("SPH_CELL_ORDER", None)
            else:
                os.environ["SPH_CELL_ORDER"] = old
    return cfg, sim


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(float(np.abs(b).max()), 1e-30))


def quantiles(a, b, q=(0.5, 0.99)):
    e = np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(1)) / max(float(np.abs(b).max()), 1e-30)
    return [float(v) for v in np.quantile(e, q)]


@pytest.mark.parametrize("scene", ["dfsph_small", "dfsph_dam_x"])
def test_relaxed_is_active_and_exact_is_not(scene):
    _, rx = make(scene, nat.ARITH_RELAXED)
    _, ex = make(scene, nat.ARITH_EXACT)
    rx.step_dfsph(1); ex.step_dfsph(1)
    assert rx.scalar(nat.S_ARITH_RELAXED) == 1.0 and ex.scalar(nat.S_ARITH_RELAXED) == 0.0
    # the same scene in the reference's cell order (quad sweeps, unstaged): since round 4 the exact sweeps with the relaxed kernel functions (KF<true>)
    _, small = make(scene, nat.ARITH_RELAXED, morton=False)
    small.step_dfsph(1)
    assert small.scalar(nat.S_ARITH_RELAXED) == 1.0
    assert rel(small.download(nat.F_POS), ex.download(nat.F_POS)) <= 1e-5 and not np.array_equal(small.download(nat.F_VEL), ex.download(nat.F_VEL))
    for s in (rx, ex, small):
        s.close()


@pytest.mark.parametrize("scene,steps", [("dfsph_small", 40), ("dfsph_dam_x", 120), ("dfsph_tiny_clamp", 30), ("breaking_dam_30k_dfsph", 30)])
def test_relaxed_density_and_alpha_per_particle(scene, steps):
    """D1 in the relaxed arithmetic (k_density_rx: W and grad W from one v_rsq_f32, the walls' share from the per-step wall sums, a tail mask
    for the particle's own padding entry) against the exact kernel on the SAME positions: rho and alpha of every particle within a few
    ulp -- walls, clamp walls, ragged cells after `steps` steps.  (The first-steps test below sees D1 only through what the solver makes of it.)"""
    cfg, ex = make(scene, nat.ARITH_EXACT)
    _, rx = make(scene, nat.ARITH_RELAXED)
    ex.step_dfsph(steps)
    for f in (nat.F_POS, nat.F_VEL, nat.F_WARM_K):
        rx.upload(f, ex.download(f))
    ex.compute_density(); rx.compute_density()
    assert rx.scalar(nat.S_ARITH_RELAXED) == 1.0
    for f, tol in ((nat.F_RHO, 2e-6), (nat.F_ALPHA, 2e-5)):
        a, b = rx.download(f).astype(np.float64), ex.download(f).astype(np.float64)
        err = np.abs(a - b) / np.maximum(np.abs(b), 1e-30 if f == nat.F_RHO else float(np.abs(b).max()) * 1e-3)
        print("%s after %d steps: field %d max rel err %.2e, median %.2e" % (scene, steps, f, err.max(), np.median(err)))
        assert err.max() < tol, (scene, f, float(err.max()))
    rx.close(); ex.close()


@pytest.mark.parametrize("scene", ["dfsph_small", "dfsph_dam_x", "dfsph_tiny_clamp", "breaking_dam_30k_dfsph",
                                   "dfsph_small:quad", "breaking_dam_30k_dfsph:quad", "dfsph_tiny_wall:quad", "dfsph_small:plain", "breaking_dam_30k_dfsph:plain"])
def test_relaxed_first_steps_within_1e5_of_the_oracle(scene, monkeypatch):
    """Five steps from rest.  Positions: within 1e-5 (max norm) of the canonical oracle -- or within 2x what seeded legal executions of
    the oracle differ from it, once THEY exceed 1e-5 (config 1: step 5) -- iteration counts equal.  Velocities cannot be
    held to a fixed 1e-5 even here -- two legal executions of the REFERENCE are 1e-3 apart (max norm) at step 3 and 4e-4 in the median at
    step 5 (profiles/r03/envelope_*.json: the rest lattice puts six neighbours at exactly r = h, one ulp flips their membership and
    with it the `neighbour count < 20` skip) -- so they are held to what the reference does to itself: per-particle median and 99 %
    quantile within 2x those of a seeded legal execution of the oracle, step by step.
    `:quad` / `:plain`: the scene in the reference's cell order -- the unstaged sweeps (four lanes per particle / one) with the relaxed kernel
    functions KF<true> (round 4); the others on the Morton curve: the staged tolerance-grade kernels."""
    scene, _, mode = scene.partition(":")
    if mode == "plain":
        monkeypatch.setenv("SPH_QUAD", "0")
    cfg, rx = make(scene, nat.ARITH_RELAXED, morton=not mode)

    canon = orc.Oracle(cfg, num_threads=8)
    legal = []
    for seed in (5, 17):
        o = orc.Oracle(cfg, num_threads=8)
        o.set_schedule(seed, 1)
        legal.append(o)
    for s in range(5):
        st = rx.step_dfsph(1)
        canon.step_dfsph(1, 100)
        for o in legal:
            o.step_dfsph(1, 100)
        assert (st.n_div, st.n_dens) == (canon.last_stats.n_div, canon.last_stats.n_dens), s
        cp, cv = canon.get(orc.F_POS), canon.get(orc.F_VEL)
        ep = rel(rx.download(nat.F_POS), cp)
        qv = quantiles(rx.download(nat.F_VEL), cv)
        lv = [max(v) for v in zip(*[quantiles(o.get(orc.F_VEL), cv) for o in legal])]
        print("%s step %d: pos max-norm %.2e (legal schedules %.2e), vel q50 %.2e q99 %.2e (legal %.2e %.2e)" % (
            scene, s + 1, ep, max(rel(o.get(orc.F_POS), cp) for o in legal), qv[0], qv[1], lv[0], lv[1]))
        lp = max(rel(o.get(orc.F_POS), cp) for o in legal)
        assert ep <= max(1e-5, 2.0 * lp), (s, ep, lp)          # 1e-5 while the reference itself keeps it (4-6 steps), then the envelope
        assert qv[0] <= max(1e-5, 2.0 * lv[0]) and qv[1] <= max(1e-5, 2.0 * lv[1]), (s, qv, lv)      # north_star's 1e-5, or the envelope once it is wider
    rx.close(); canon.close()
    for o in legal:
        o.close()


def test_relaxed_stays_inside_the_reference_envelope_100_steps():
    """dfsph_config_backup's geometry (SURVEY.md 8c iii), 100 steps: the relaxed kernels against the canonical oracle, next to two
    seeded LEGAL executions of the oracle (racy cell-list order, f32 atomic means) against the same canonical run.  Relaxed must not
    be further out than the reference is from itself: per-particle median and 99 % quantile of the position / velocity deviation
    within 2x the larger of the two legal runs' (they differ between seeds by about that)."""
    scene = "dfsph_small"
    cfg, rx = make(scene, nat.ARITH_RELAXED)
    canon = orc.Oracle(cfg, num_threads=8)
    legal = []
    for seed in (11, 23):
        o = orc.Oracle(cfg, num_threads=8)
        o.set_schedule(seed, 1)
        legal.append(o)
    for s in range(100):
        rx.step_dfsph(1)
        canon.step_dfsph(1, 100)
        for o in legal:
            o.step_dfsph(1, 100)
        if s + 1 in (10, 50, 100):
            cp, cv = canon.get(orc.F_POS), canon.get(orc.F_VEL)
            rq = quantiles(rx.download(nat.F_POS), cp) + quantiles(rx.download(nat.F_VEL), cv)
            lq = [max(v) for v in zip(*[quantiles(o.get(orc.F_POS), cp) + quantiles(o.get(orc.F_VEL), cv) for o in legal])]
            print("step %3d  relaxed pos q50 %.2e q99 %.2e vel q50 %.2e q99 %.2e | legal schedules pos %.2e %.2e vel %.2e %.2e | max-norm pos relaxed %.2e legal %.2e" % (
                (s + 1,) + tuple(rq) + tuple(lq) + (rel(rx.download(nat.F_POS), cp), max(rel(o.get(orc.F_POS), cp) for o in legal))))
            for a, b in zip(rq, lq):
                assert a <= 2.0 * b + 1e-7, (s + 1, rq, lq)
    pos = rx.download(nat.F_POS)
    assert np.isfinite(pos).all() and pos.min() >= 0.0
    rx.close(); canon.close()
    for o in legal:
        o.close()


def test_relaxed_dfsph_1m_20_steps_from_the_timed_phase():
    """Config 3 where bench.py times it: 55 exact steps, then the state goes to an exact and to a relaxed handle and both run 20 steps.
    Same physics: iteration counts within 2 of each other step for step, densities equally converged, per-particle median deviation
    under 1e-5 after 20 steps, nothing lost."""
    cfg = scenes.get("dfsph_1m")
    ex = nat.Simulation(nat.config_from_dict(cfg))
    ex.step_dfsph(55)
    rx = nat.Simulation(nat.config_from_dict(cfg, arith=nat.ARITH_RELAXED))
    for f in (nat.F_POS, nat.F_VEL, nat.F_WARM_K):
        rx.upload(f, ex.download(f))
    rx.set_dt(ex.scalar(nat.S_DELTA_TIME))
    diffs = []
    for s in range(20):
        a, b = ex.step_dfsph(1), rx.step_dfsph(1)
        assert rx.scalar(nat.S_ARITH_RELAXED) == 1.0
        assert b.lost == 0 and b.capped == 0
        assert abs(a.n_dens - b.n_dens) <= 2 and abs(a.n_div - b.n_div) <= 2, (s, a.n_div, a.n_dens, b.n_div, b.n_dens)
        assert abs(a.dens_err - b.dens_err) <= 0.02 and abs(a.dt - b.dt) <= 1e-4 * a.dt, (s, a.dens_err, b.dens_err, a.dt, b.dt)
        diffs.append((a.n_dens, b.n_dens))
    qp = quantiles(rx.download(nat.F_POS), ex.download(nat.F_POS), (0.5, 0.99, 0.999))
    qv = quantiles(rx.download(nat.F_VEL), ex.download(nat.F_VEL), (0.5, 0.99, 0.999))
    print("dfsph_1m, 20 steps from step 55: (n_dens exact, relaxed) %s; pos q50/q99/q999 %.2e %.2e %.2e; vel %.2e %.2e %.2e" % ((diffs,) + tuple(qp) + tuple(qv)))
    assert qp[0] <= 1e-5 and qp[1] <= 1e-4
    ex.close(); rx.close()


# ---- WCSPH under the relaxed arithmetic: Verlet lists, two kernels per step --------------------------------------------
# WCSPH is the solver whose own nondeterminism envelope is tiny (4e-8 after 200 steps), so here north_star's bar --
# positions and velocities within 1e-5 (max norm) of the reference after N steps -- is tested DIRECTLY against the oracle.

def wcsph_pair(scene):
    cfg = scenes.get(scene)
    rx = nat.Simulation(nat.config_from_dict(cfg, arith=nat.ARITH_RELAXED))
    return cfg, rx


@pytest.mark.parametrize("scene,steps", [("breaking_dam_30k_wcsph", 200), ("wcsph_small", 400), ("wcsph_tiny_wall", 300), ("wcsph_tiny_clamp", 300),
                                         ("wcsph_dam_x", 1500)])
def test_relaxed_wcsph_within_1e5_of_the_oracle(scene, steps):
    """Config 1 (29 k particles) 200 steps from rest, and smaller scenes far longer (wcsph_dam_x: the column collapses and runs along the box,
    the lists are rebuilt many times): max-norm deviation from the canonical oracle <= 1e-5 in positions AND velocities at several points
    of the run -- or, where a seeded LEGAL execution of the oracle (racy cell-list order: what the reference does to itself) is further than
    that from the canonical one, within 3x of the larger of two of them: a column of a few hundred particles pressed against a wall by the stiff Tait equation is
    not as well conditioned as the free dam.  The run must really have reused lists (fewer builds than steps)."""
    cfg, rx = wcsph_pair(scene)
    o = orc.Oracle(cfg, num_threads=8)
    legal = []
    for seed in (7, 19):
        lo = orc.Oracle(cfg, num_threads=8)
        lo.set_schedule(seed, 1)
        legal.append(lo)
    marks = sorted({max(1, steps // 4), steps // 2, steps})
    done = 0
    for m in marks:
        rx.step_wcsph(m - done); o.step_wcsph(m - done)
        for lo in legal:
            lo.step_wcsph(m - done)
        done = m
        ep, ev = rel(rx.download(nat.F_POS), o.get(orc.F_POS)), rel(rx.download(nat.F_VEL), o.get(orc.F_VEL))
        lp = max(rel(lo.get(orc.F_POS), o.get(orc.F_POS)) for lo in legal)
        lv = max(rel(lo.get(orc.F_VEL), o.get(orc.F_VEL)) for lo in legal)
        print("%s step %d: pos max-norm %.2e vel max-norm %.2e (legal schedules of the reference: %.2e %.2e)" % (scene, m, ep, ev, lp, lv))
        assert ep <= max(1e-5, 3.0 * lp) and ev <= max(1e-5, 3.0 * lv), (scene, m, ep, ev, lp, lv)
        if scene == "breaking_dam_30k_wcsph":
            assert ep <= 1e-5 and ev <= 1e-5          # BASELINE config 1: north_star's bar as stated
    assert rx.scalar(nat.S_ARITH_RELAXED) == 1.0
    builds = rx.scalar(nat.S_VERLET_BUILDS)
    print("%s: %d list builds in %d steps" % (scene, builds, steps))
    assert 1 <= builds < steps
    rx.close(); o.close()
    for lo in legal:
        lo.close()


def test_relaxed_wcsph_250k_steps_151_to_155():
    """Config 2 in the phase its bench line times: 150 exact steps on the device, then the state goes to a relaxed handle and to the oracle,
    both run steps 151-155; <= 1e-5 max norm after each."""
    cfg = scenes.get("wcsph_250k")
    ex = nat.Simulation(nat.config_from_dict(cfg))
    ex.step_wcsph(150)
    pos, vel = ex.download(nat.F_POS), ex.download(nat.F_VEL)
    ex.close()
    rx = nat.Simulation(nat.config_from_dict(cfg, arith=nat.ARITH_RELAXED))
    rx.upload(nat.F_POS, pos); rx.upload(nat.F_VEL, vel)
    o = orc.Oracle(cfg, num_threads=16)
    o.set(orc.F_POS, pos); o.set(orc.F_VEL, vel)
    for s in range(5):
        rx.step_wcsph(1); o.step_wcsph(1)
        ep, ev = rel(rx.download(nat.F_POS), o.get(orc.F_POS)), rel(rx.download(nat.F_VEL), o.get(orc.F_VEL))
        print("wcsph_250k step %d: pos max-norm %.2e vel max-norm %.2e" % (151 + s, ep, ev))
        assert ep <= 1e-5 and ev <= 1e-5, (s, ep, ev)
    assert rx.scalar(nat.S_ARITH_RELAXED) == 1.0
    rx.close(); o.close()


def test_relaxed_wcsph_list_reuse_is_only_a_schedule():
    """SPH_VERLET_SKIN=0 rebuilds the lists every step (a zero skin: any motion counts as moved): the same relaxed kernels on canonical
    lists.  The reused-list run stays within 1e-6 of it over 300 steps -- what changes between two builds is the ORDER of the sums and which
    pairs beyond h are visited (they contribute exactly 0)."""
    cfg = scenes.get("wcsph_small")
    a = nat.Simulation(nat.config_from_dict(cfg, arith=nat.ARITH_RELAXED))
    os.environ["SPH_VERLET_SKIN"] = "0"
    try:
        b = nat.Simulation(nat.config_from_dict(cfg, arith=nat.ARITH_RELAXED))
    finally:
        os.environ.pop("SPH_VERLET_SKIN", None)
    a.step_wcsph(300); b.step_wcsph(300)
    print("builds", a.scalar(nat.S_VERLET_BUILDS), b.scalar(nat.S_VERLET_BUILDS), rel(a.download(nat.F_POS), b.download(nat.F_POS)), rel(a.download(nat.F_VEL), b.download(nat.F_VEL)))
    assert b.scalar(nat.S_VERLET_BUILDS) == 300 and a.scalar(nat.S_VERLET_BUILDS) < 150
    assert rel(a.download(nat.F_POS), b.download(nat.F_POS)) <= 1e-6 and rel(a.download(nat.F_VEL), b.download(nat.F_VEL)) <= 1e-5
    a.close(); b.close()


def test_arith_through_the_mirror_api(capsys):
    """ParticleSystem(config, arith=...), <name>_solver(ps, config, arith=...) and run.py --arith reach SphConfig.arith (VERDICT r3 missing #3)."""
    from cfd_taichi_amd import ParticleSystem, wcsph_solver
    cfg = scenes.get("wcsph_tiny_wall")
    ps = ParticleSystem(cfg, arith="relaxed")
    sol = wcsph_solver(ps, cfg)
    sol.step(3)
    assert sol.arith == "relaxed" and sol._sim.scalar(nat.S_ARITH_RELAXED) == 1.0
    ps2 = ParticleSystem(cfg)
    sol2 = wcsph_solver(ps2, cfg)
    sol2.step(3)
    assert sol2.arith == "exact" and sol2._sim.scalar(nat.S_ARITH_RELAXED) == 0.0
    sol3 = wcsph_solver(ParticleSystem(cfg), cfg, arith="relaxed")          # the solver may ask for it too: the handle is rebuilt
    sol3.step(3)
    assert sol3.arith == "relaxed" and sol3._sim.scalar(nat.S_ARITH_RELAXED) == 1.0
    assert rel(sol3.ps.fluid_particles.pos.to_numpy(), sol2.ps.fluid_particles.pos.to_numpy()) <= 1e-6
    cfg4 = scenes.get("wcsph_tiny_wall")
    cfg4["solver"]["arith"] = "relaxed"                                      # ... or the config
    assert ParticleSystem(cfg4).arith == nat.ARITH_RELAXED
    with pytest.raises(ValueError):
        ParticleSystem(cfg, arith="fast")


@pytest.mark.parametrize("scene,steps", [("dfsph_rigid_small", 40), ("dfsph_rigid_tilted", 20)])
def test_relaxed_next_to_a_rigid_body(scene, steps, monkeypatch):
    """Round 4: SPH_ARITH_RELAXED on a handle with a coupled body.  The tolerance-grade sweeps take the workgroups without a rigid sample in reach
    (16-bit lists), the exact RIGID sweeps the shell around the body -- two launches per sweep over one tile order (rx_split).  First steps against
    the ORACLE (iteration counts equal, positions within 1e-5 or the reference's own envelope, the force on the body within 1e-3 of its magnitude),
    then coupled steps next to the exact kernels: same physics (iteration counts within 2 up to a one-step shift of the loop's onset, body within 2.5e-4, nothing lost)."""
    from cfd_taichi_amd import mesh
    cfg = scenes.get(scene)
    rg = mesh.rigid_from_config(cfg)
    monkeypatch.setenv("SPH_CELL_ORDER", "morton")
    rx = nat.Simulation(nat.config_from_dict(cfg, arith=nat.ARITH_RELAXED), rigid=rg)
    ex = nat.Simulation(nat.config_from_dict(cfg), rigid=rg)
    o = orc.Oracle(cfg, num_threads=8, rigid=rg)
    legal = orc.Oracle(cfg, num_threads=8, rigid=rg)
    legal.set_schedule(3, 1)
    for s in range(3):
        st = rx.step_dfsph(1); ex.step_dfsph(1)
        o.step_dfsph(1, 100); legal.step_dfsph(1, 100)
        assert rx.scalar(nat.S_ARITH_RELAXED) == 1.0 and ex.scalar(nat.S_ARITH_RELAXED) == 0.0
        assert (st.n_div, st.n_dens) == (o.last_stats.n_div, o.last_stats.n_dens), s
        ep, lp = rel(rx.download(nat.F_POS), o.get(orc.F_POS)), rel(legal.get(orc.F_POS), o.get(orc.F_POS))
        fo = o.get(orc.F_RIGID_FORCE)
        print("%s step %d: pos max-norm %.2e (legal schedule %.2e)" % (scene, s + 1, ep, lp))
        assert ep <= max(1e-5, 2.0 * lp), (s, ep, lp)
        if fo is not None:
            fr = rx.download(nat.F_RIGID_FORCE, nat.SPECIES_RIGID)
            assert np.abs(fr.sum(0) - fo.sum(0)).max() <= 1e-3 * max(float(np.abs(fo.sum(0)).max()), 1e-6), (s, fr.sum(0), fo.sum(0))
        for sim in (rx, ex):
            sim.rigid_step()
        o.rigid_step(); legal.rigid_step()
    seq_a, seq_b = [], []
    for s in range(3, steps):
        a, b = ex.step_dfsph(1), rx.step_dfsph(1)
        o.step_dfsph(1, 100); legal.step_dfsph(1, 100)
        assert b.lost == 0 and abs(a.n_div - b.n_div) <= 2, (s, a.n_div, b.n_div)
        seq_a.append(a.n_dens); seq_b.append(b.n_dens)
        ex.rigid_step(); rx.rigid_step(); o.rigid_step(); legal.rigid_step()
    # the density loop wakes up when the falling body starts to squeeze the fluid under it; the two arithmetics may see that one step apart
    for k, nb in enumerate(seq_b):
        assert min(abs(nb - na) for na in seq_a[max(0, k - 1):k + 2]) <= 2, (k + 3, seq_a, seq_b)
    ca, cb = np.asarray(ex.rigid_scalars()["centroid"]), np.asarray(rx.rigid_scalars()["centroid"])
    cl = np.asarray(legal.rigid_scalars()["centroid"])
    q = quantiles(rx.download(nat.F_POS), o.get(orc.F_POS), (0.5, 0.99))
    ql = quantiles(legal.get(orc.F_POS), o.get(orc.F_POS), (0.5, 0.99))
    print("%s after %d coupled steps: centroid exact %s relaxed %s legal schedule %s; fluid pos vs the canonical oracle q50 %.2e q99 %.2e (legal schedule: %.2e %.2e)" % (
        scene, steps, ca, cb, cl, q[0], q[1], ql[0], ql[1]))
    assert np.array_equal(ex.download(nat.F_POS), o.get(orc.F_POS))                      # (the exact handle IS the canonical execution)
    assert np.abs(ca - cb).max() <= max(2.5e-4, 3.0 * float(np.abs(ca - cl).max()))      # the body: within what a legal schedule of the reference does to it
    assert q[0] <= 3.0 * ql[0] + 1e-6 and q[1] <= 3.0 * ql[1] + 1e-6 and np.isfinite(rx.download(nat.F_POS)).all()
    for sim in (rx, ex):
        sim.close()
    o.close(); legal.close()


# ---- PCISPH / IISPH under the relaxed arithmetic (round 4: the sweeps take the kernel functions KF<true>, sph_device.h) --------------------------

@pytest.mark.parametrize("scene,steps,morton", [("breaking_dam_30k_pcisph", 12, True), ("breaking_dam_30k_iisph", 40, True), ("dfsph_tiny_wall_pcisph", 60, False),
                                                ("dfsph_tiny_wall_iisph", 60, False), ("iisph_config_backup", 60, True), ("pcisph_config_backup", 40, True)])
def test_relaxed_pressure_solvers_stay_inside_the_reference_envelope(scene, steps, morton, monkeypatch):
    """SPH_ARITH_RELAXED on pcisph / iisph handles: staged sweeps (Morton curve) and the plain one-lane sweeps (SPH_QUAD=0 for the small scenes).
    Against the canonical oracle next to two seeded legal executions of the oracle: positions and velocities within max(1e-5, 3x the legal
    schedules' own deviation) at several points of the run, the pressure loop's iteration count within 2 (or 10 %) of the oracle's."""
    cfg = scenes.get(scene)
    if morton:
        monkeypatch.setenv("SPH_CELL_ORDER", "morton")
    else:
        monkeypatch.setenv("SPH_QUAD", "0")
    rx = nat.Simulation(nat.config_from_dict(cfg, arith=nat.ARITH_RELAXED))
    o = orc.Oracle(cfg, num_threads=8)
    legal = []
    for seed in (7, 19):
        lo = orc.Oracle(cfg, num_threads=8)
        lo.set_schedule(seed, 1)
        legal.append(lo)
    pci = cfg["solver"]["name"] == "pcisph"
    marks = sorted({max(1, steps // 4), steps // 2, steps})
    for s in range(steps):
        st = rx.step(1)
        (o.step_pcisph if pci else o.step_iisph)(1)
        for lo in legal:
            (lo.step_pcisph if pci else lo.step_iisph)(1)
        assert rx.scalar(nat.S_ARITH_RELAXED) == 1.0
        no = o.last_stats.n_dens
        assert abs(st.n_dens - no) <= max(2, no // 10), (s, st.n_dens, no)
        if s + 1 in marks:
            ep, ev = rel(rx.download(nat.F_POS), o.get(orc.F_POS)), rel(rx.download(nat.F_VEL), o.get(orc.F_VEL))
            lp = max(rel(lo.get(orc.F_POS), o.get(orc.F_POS)) for lo in legal)
            lv = max(rel(lo.get(orc.F_VEL), o.get(orc.F_VEL)) for lo in legal)
            print("%s step %d: pos max-norm %.2e vel max-norm %.2e (legal schedules of the reference: %.2e %.2e), %d / %d pressure iterations" % (scene, s + 1, ep, ev, lp, lv, st.n_dens, no))
            assert ep <= max(1e-5, 3.0 * lp) and ev <= max(1e-5, 3.0 * lv), (scene, s + 1, ep, ev, lp, lv)
    rx.close(); o.close()
    for lo in legal:
        lo.close()
