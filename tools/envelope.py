#!/usr/bin/env python3
"""How far do two LEGAL executions of the reference drift apart?  (CPU only, uses the oracle.)

The reference is not deterministic: the cell lists are appended from a parallel loop (ParticleSystem.py:388-397), so the order of every
neighbour sum is whatever the thread schedule produced, and the DFSPH residual means are f32 atomics (dfsph_solver.py:139-141, 275-279).
oracle.set_schedule(seed, chunk) draws one such execution (seeded order inside every cell, redrawn at every grid rebuild; f32 means with
`chunk` particles per thread-local partial).  This tool runs the canonical oracle (single-thread order, f64 means), the f64 oracle and
S seeded executions per chunk size on one scene and reports, per step:

  spread_pos / spread_vel   max over seeds of  max|x_seed - x_canonical| / max|x_canonical|      (the norm of north_star's 1e-5 bar)
  pair_pos / pair_vel       max over seed pairs of the same quantity (two legal runs against each other)
  f64_pos / f64_vel         canonical f32 run against the f64 run (what f32 rounding alone costs)
  flips                     seeds whose (n_div, n_dens) of the step differ from the canonical run's

    python tools/envelope.py --scene breaking_dam_30k_dfsph --steps 60 --seeds 8 --out profiles/r03/envelope_c1_dfsph.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cfd_taichi_amd import scenes          # noqa: E402
from oracle import oracle as orc           # noqa: E402


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(float(np.abs(b).max()), 1e-30))


def rel_q(a, b, q):
    """per-particle |a_i - b_i| (Euclidean) / max|b|: the q-quantiles over the particles (the max norm is set by a handful of particles
    whose discrete gates flipped -- list membership at r = h, the `neighbour count < 20` skip, max(., 0) -- the quantiles say what the rest does)"""
    e = np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(1)) / max(float(np.abs(b).max()), 1e-30)
    return [float(v) for v in np.quantile(e, q)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", required=True)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--chunks", default="1,2048", help="particles per thread-local partial of the f32 means (1 = one atomic per particle)")
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--every", type=int, default=1, help="compare every k-th step")
    ap.add_argument("--dense", type=int, default=20, help="... and every step up to this one")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()

    cfg = scenes.get(args.scene)
    kind = cfg["solver"]["name"]
    assert kind in ("wcsph", "dfsph"), "the envelope is defined for the solvers of BASELINE.json (wcsph, dfsph)"
    chunks = [int(c) for c in args.chunks.split(",")] if kind == "dfsph" else [1]

    def make(precision="f32"):
        return orc.Oracle(cfg, num_threads=args.threads, precision=precision)

    canon, f64 = make(), make("f64")
    runs = []
    for ch in chunks:
        for s in range(args.seeds):
            o = make()
            o.set_schedule(1000003 * (s + 1) + ch, ch)
            runs.append((ch, s, o))

    def step(o):
        if kind == "dfsph":
            o.step_dfsph(1, 100)
            return (o.last_stats.n_div, o.last_stats.n_dens)
        o.step_wcsph(1)
        return (0, 0)

    rows = []
    for k in range(1, args.steps + 1):
        c_it = step(canon)
        step(f64)
        its = [step(o) for _, _, o in runs]
        if k > args.dense and k % args.every and k != args.steps:
            continue
        cp, cv = canon.get(orc.F_POS), canon.get(orc.F_VEL)
        row = {"step": k, "canonical_iters": list(c_it), "f64_pos": rel(f64.get(orc.F_POS), cp), "f64_vel": rel(f64.get(orc.F_VEL), cv)}
        for ch in chunks:
            sel = [(o, it) for (c, _, o), it in zip(runs, its) if c == ch]
            P = [o.get(orc.F_POS) for o, _ in sel]
            V = [o.get(orc.F_VEL) for o, _ in sel]
            key = "chunk%d" % ch
            row[key] = {
                "spread_pos": max(rel(p, cp) for p in P), "spread_vel": max(rel(v, cv) for v in V),
                "pair_pos": max(rel(P[a], P[b]) for a in range(len(P)) for b in range(a)) if len(P) > 1 else 0.0,
                "pair_vel": max(rel(V[a], V[b]) for a in range(len(V)) for b in range(a)) if len(V) > 1 else 0.0,
                "flips": sum(1 for _, it in sel if tuple(it) != tuple(c_it)), "iters": [list(it) for _, it in sel],
                # seed 0 against the canonical run, per-particle quantiles (median, 99 %, 99.9 %)
                "pos_q50_q99_q999": rel_q(P[0], cp, [0.5, 0.99, 0.999]), "vel_q50_q99_q999": rel_q(V[0], cv, [0.5, 0.99, 0.999]),
            }
        rows.append(row)
        r0 = row["chunk%d" % chunks[0]]
        print("step %4d  canonical %s  spread pos %.2e vel %.2e  pair pos %.2e  f64 pos %.2e  flips %d/%d  pos q50/q99 %.1e %.1e  vel q50/q99 %.1e %.1e" % (
            k, c_it, r0["spread_pos"], r0["spread_vel"], r0["pair_pos"], row["f64_pos"], r0["flips"], args.seeds,
            r0["pos_q50_q99_q999"][0], r0["pos_q50_q99_q999"][1], r0["vel_q50_q99_q999"][0], r0["vel_q50_q99_q999"][1]), flush=True)

    def first_over(key, sub, bar=1e-5):
        for r in rows:
            v = r[sub][key] if sub else r[key]
            if v > bar:
                return r["step"]
        return None

    summary = {"scene": args.scene, "solver": kind, "particles": canon.N, "steps": args.steps, "seeds": args.seeds, "chunks": chunks,
               "norm": "max|a - b| / max|b| over all particles and components (the norm of smoke() and of north_star's 1e-5 bar)",
               "first_step_f64_pos_over_1e-5": first_over("f64_pos", None)}
    for ch in chunks:
        key = "chunk%d" % ch
        summary[key] = {"first_step_spread_pos_over_1e-5": first_over("spread_pos", key), "first_step_spread_vel_over_1e-5": first_over("spread_vel", key),
                        "first_step_with_iteration_flip": next((r["step"] for r in rows if r[key]["flips"]), None),
                        "final_spread_pos": rows[-1][key]["spread_pos"], "final_spread_vel": rows[-1][key]["spread_vel"],
                        "final_pair_pos": rows[-1][key]["pair_pos"]}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump({"summary": summary, "rows": rows}, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
