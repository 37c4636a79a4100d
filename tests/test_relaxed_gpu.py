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
    # a handle the relaxed sweeps do not cover (small scene in the reference's cell order: quad sweeps) runs the exact ones
    _, small = make(scene, nat.ARITH_RELAXED, morton=False)
    small.step_dfsph(1)
    assert small.scalar(nat.S_ARITH_RELAXED) == 0.0
    assert np.array_equal(small.download(nat.F_POS), ex.download(nat.F_POS))
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


@pytest.mark.parametrize("scene", ["dfsph_small", "dfsph_dam_x", "dfsph_tiny_clamp", "breaking_dam_30k_dfsph"])
def test_relaxed_first_steps_within_1e5_of_the_oracle(scene):
    """Five steps from rest.  Positions: within 1e-5 (max norm) of the canonical oracle -- or within 2x what seeded legal executions of
    the oracle differ from it, once THEY exceed 1e-5 (config 1: step 5) -- iteration counts equal.  Velocities cannot be
    held to a fixed 1e-5 even here -- two legal executions of the REFERENCE are 1e-3 apart (max norm) at step 3 and 4e-4 in the median at
    step 5 (profiles/r03/envelope_*.json: the rest lattice puts six neighbours at exactly r = h, one ulp flips their membership and
    with it the `neighbour count < 20` skip) -- so they are held to what the reference does to itself: per-particle median and 99 %
    quantile within 2x those of a seeded legal execution of the oracle, step by step."""
    cfg, rx = make(scene, nat.ARITH_RELAXED)
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
