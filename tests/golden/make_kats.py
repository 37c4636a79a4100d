"""Generates tests/golden/kats.json: analytic known-answer values for the SPH path, computed in
plain numpy f64 from the formulas as written in the reference (solver_base.py:76-103,
wcsph_solver.py:86-90, ParticleSystem.py:83-137, dfsph_solver.py:32-51).  Nothing from the
reference is imported or executed (its runtime, taichi, is not installed); these values pin the
oracle, which is otherwise 'parity unpinned'.   Run:  python tests/golden/make_kats.py
"""
import itertools
import json
import math
import os

import numpy as np

R = 0.025
D = 2 * R
H = 4 * R
M = 1000 * R ** 3 * 8
RHO0 = 1000.0


def cubic_kernel(r, h):                       # solver_base.py:76-88
    q = r / h
    k = 8 / (math.pi * h ** 3)
    if 0 <= q <= 0.5:
        return k * (6 * (q ** 3 - q ** 2) + 1)
    if 0.5 < q <= 1:
        return 2 * k * (1 - q) ** 3
    return 0.0


def cubic_kernel_derivative(rv, h):           # solver_base.py:90-103 (factor 6 as written)
    rv = np.asarray(rv, dtype=np.float64)
    rn = float(np.linalg.norm(rv))
    q = rn / h
    k = 48 / (math.pi * h ** 3)
    if 1e-5 < q <= 0.5:
        return k * 6 * (3 * q * q - 2 * q) * rv / (h * rn)
    if 0.5 < q <= 1:
        return -k * 6 * (1 - q) ** 2 * rv / (h * rn)
    return np.zeros(3)


def sizes(box_max, water_size):               # ParticleSystem.py:85-86,100-101,129-137
    n = int(water_size[0] / D * water_size[1] / D * water_size[2] / D)
    x_cnt = int(box_max[0] / D + 1)
    z_cnt = int(box_max[2] / D + 1)
    ring = x_cnt * z_cnt - (x_cnt - 2) * (z_cnt - 2)
    layer = int(math.ceil((box_max[1] - D) / D))
    nb = layer * ring + 2 * x_cnt * z_cnt
    grid = [int(math.ceil(b / H)) + 1 for b in box_max]
    return {"N": n, "Nb": nb, "grid": grid, "C": grid[0] * grid[1] * grid[2]}


def main():
    kats = {"particle_m": M, "W0": cubic_kernel(0.0, H), "Wd": cubic_kernel(D, H), "Wh": cubic_kernel(H, H)}
    # integral of W over its support (spherical shells)
    rr = np.linspace(0.0, H, 200001)
    w = np.array([cubic_kernel(x, H) for x in rr])
    kats["W_integral"] = float(np.trapezoid(4 * math.pi * rr ** 2 * w, rr))
    kats["sizes"] = {
        "breaking_dam_30k": sizes([5.0, 3.0, 1.5], [1.0, 2.8, 1.3]),
        "dfsph_config_backup": sizes([1.5, 3.0, 1.5], [0.7, 1.5, 0.7]),
        "coupling_demo_fluid": sizes([5.0, 7.0, 2.5], [1.5, 2.0, 2.3]),
        "wcsph_250k": sizes([10.0, 6.0, 2.7], [2.5, 5.0, 2.5]),
        "dfsph_1m": sizes([16.0, 7.0, 5.2], [5.0, 5.0, 5.0]),
        "dfsph_10m": sizes([40.0, 15.0, 10.2], [10.0, 12.5, 10.0]),
    }
    # interior particle of the rest lattice (spacing d): self excluded, r == h included
    offs = [np.array(o, dtype=np.float64) * D for o in itertools.product(range(-3, 4), repeat=3) if o != (0, 0, 0)]
    nb = [o for o in offs if np.linalg.norm(o) <= H * (1 + 1e-12)]
    rho = 0.001 + sum(M * cubic_kernel(float(np.linalg.norm(o)), H) for o in nb)       # solver_base.py:44,62
    grads = [M * cubic_kernel_derivative(-o, H) for o in nb]
    s = np.sum(grads, axis=0)
    q = float(sum(g.dot(g) for g in grads))
    kats["lattice_interior"] = {
        "neighbors_le_h": len(nb),
        "neighbors_lt_h": len([o for o in nb if np.linalg.norm(o) < H * (1 - 1e-12)]),
        "rho": rho,
        "rho_with_self": rho + M * cubic_kernel(0.0, H),
        "sum_sq_grad": q,
        "norm_sum_grad": float(np.linalg.norm(s)),
        "alpha": rho / (float(s.dot(s)) + q),                                           # dfsph_solver.py:45-51
    }
    # interior particle of a flat single-layer wall (spacing d): V_b = 1 / sum_{k != b} W    ParticleSystem.py:309-320
    woffs = [np.array([i, 0, j], dtype=np.float64) * D for i in range(-3, 4) for j in range(-3, 4) if (i, j) != (0, 0)]
    wn = [o for o in woffs if np.linalg.norm(o) <= H * (1 + 1e-12)]
    sw = sum(cubic_kernel(float(np.linalg.norm(o)), H) for o in wn)
    kats["flat_wall"] = {"neighbors": len(wn), "sum_W": sw, "volume": 1.0 / sw}
    # Tait EOS                                                                           wcsph_solver.py:86-90
    kats["tait"] = {"p_1010": 70000 * ((1010 / RHO0) ** 7 - 1.0), "p_le_1000": 0.0}
    # first DFSPH step from rest: interior v* = (0, -dt*g/m, 0)  (gravity not scaled by m, dfsph_solver.py:96,102)
    kats["dfsph_first_step"] = {"gravity_over_m": 9.8 / M, "dt_after_step1": 1e-3}
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kats.json")
    with open(out, "w") as f:
        json.dump(kats, f, indent=2, sort_keys=True)
        f.write("\n")
    print(json.dumps(kats, indent=2, sort_keys=True))


if __name__ == "__main__":
    main()
