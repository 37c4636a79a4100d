#!/usr/bin/env python3
"""Exact and relaxed handles from the same state, step by step: iteration counts, residuals, deviation quantiles.
    python tools/relaxed_vs_exact.py [scene] [pre_steps] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cfd_taichi_amd import _native as nat, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "dfsph_1m"
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 70
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
cfg = scenes.get(scene)
ex = nat.Simulation(nat.config_from_dict(cfg))
ex.step_dfsph(pre)
state = [ex.download(f) for f in (nat.F_POS, nat.F_VEL, nat.F_WARM_K)]
dt = ex.scalar(nat.S_DELTA_TIME)
rx = nat.Simulation(nat.config_from_dict(cfg, arith=nat.ARITH_RELAXED))
ex2 = nat.Simulation(nat.config_from_dict(cfg))          # a second EXACT handle from the same state: the hand-over itself must be invisible
for h in (rx, ex2):
    for f, v in zip((nat.F_POS, nat.F_VEL, nat.F_WARM_K), state):
        h.upload(f, v)
    h.set_dt(dt)
for s in range(steps):
    a, b, c = ex.step_dfsph(1), rx.step_dfsph(1), ex2.step_dfsph(1)
    if s % 5 == 0 or s == steps - 1:
        e = np.sqrt(((rx.download(nat.F_POS).astype(np.float64) - ex.download(nat.F_POS)) ** 2).sum(1)) / np.abs(ex.download(nat.F_POS)).max()
        print("step %3d exact (%2d,%2d) dens_err %.4f div_err %.3f dt %.3e | relaxed (%2d,%2d) dens_err %.4f div_err %.3f dt %.3e | exact2 (%2d,%2d) | pos q50 %.1e q99 %.1e max %.1e" % (
            pre + s + 1, a.n_div, a.n_dens, a.dens_err, a.div_err, a.dt, b.n_div, b.n_dens, b.dens_err, b.div_err, b.dt, c.n_div, c.n_dens,
            np.quantile(e, 0.5), np.quantile(e, 0.99), e.max()), flush=True)
print("exact == exact2:", np.array_equal(ex.download(nat.F_POS), ex2.download(nat.F_POS)))
