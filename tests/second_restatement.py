"""A SECOND, independent restatement of the reference's WCSPH / DFSPH step -- numpy f32, brute-force O(N^2) neighbour search.

TEST INFRASTRUCTURE ONLY (tests/test_second_restatement.py).  Purpose: a transcription check on oracle/sph_oracle.c.  The oracle and the
HIP kernels were written by the same hand from the same reading of the reference, and 344 GPU tests prove that they agree with EACH OTHER;
an error both share would pass all of them.  This file was written afterwards, from the reference's text alone (files and lines cited at
every function; /root/reference, Jukgei/CFD_Taichi @ 2024_08_07), with a different algorithm and shares no code with oracle/ or the library:

  * NO cell lists.  A neighbour of i is any j != i whose grid cell differs from i's by at most one in every axis and lies inside the
    grid, with |x_i - x_j| <= support_radius -- found by testing all N^2 (resp. N x Nb) pairs (ParticleSystem.py:447-469, 337-366 state
    the same set through the 27-cell walk).  A particle's neighbours are then ordered the way the reference's walk meets them:
    cell offset (dx, dy, dz) with dx outermost (`ti.ndrange((-1,2),(-1,2),(-1,2))`, :452), ascending particle index inside a cell
    (the single-thread append order of update_grid, :388-397).
  * every per-particle sum is taken neighbour by neighbour in that order (`ret += task(...)`, :469), vectorised over the PARTICLES only.

Arithmetic conventions (the assumptions about Taichi that SURVEY.md Appendix A lists; they are conventions, not transcriptions): f32 fields
and kernel locals; a sub-expression made of Python scalars only is evaluated in f64 and rounded to f32 where it meets a Taichi value;
x ** n for a literal integer n by binary exponentiation; vector norm = sqrt((x^2 + y^2) + z^2), dot likewise; no FMA; IEEE divide / sqrt;
kernel-scope f32 sums that Taichi turns into atomics (the residual means, dfsph_solver.py:139-149, 275-279) are taken exactly (math.fsum)
and rounded once.  No rigid body (the tiny scenes have none): the material_solid branches are not restated.
"""
import math

import numpy as np

F = np.float32


def _ipow(x, n):
    """x ** n, n a literal integer >= 1: right-to-left binary exponentiation"""
    result, base = None, x
    while n:
        if n & 1:
            result = base if result is None else result * base
        n >>= 1
        if n:
            base = base * base
    return result


def _norm(v):                     # ti.Vector.norm()
    return np.sqrt((v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2])


def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


class Scene:
    """ParticleSystem.__init__ (ParticleSystem.py:31-127): sizes, the fluid lattice, the wall particles and their volumes."""

    def __init__(self, config):
        scene, fluid = config["scene"], config["fluid"]
        self.radius = scene["particle_radius"]                                     # :80
        self.diameter = self.radius * 2                                            # :81
        self.support = 4 * self.radius                                             # :82
        self.m = 1000 * (self.radius ** 3) * 8                                     # :83
        ws, self.start = fluid["water_size"], fluid["start_pos"]
        self.N = int(ws[0] / self.diameter * ws[1] / self.diameter * ws[2] / self.diameter)      # :85-86
        self.box_max, self.box_min = scene["box_max"], scene["box_min"]
        box = [self.box_max[a] - self.box_min[a] for a in range(3)]
        # compute_boundary_particles_count :129-137
        x_cnt, z_cnt = int(box[0] / self.diameter + 1), int(box[2] / self.diameter + 1)
        bottom = x_cnt * z_cnt
        ring = x_cnt * z_cnt - (x_cnt - 2) * (z_cnt - 2)
        layers = int(math.ceil((box[1] - self.diameter) / self.diameter))
        self.Nb = layers * ring + bottom * 2
        self.grid = [int(math.ceil(box[a] / self.support)) + 1 for a in range(3)]  # :100-101
        self.C = self.grid[0] * self.grid[1] * self.grid[2]
        self.h = F(self.support)                                                   # support_radius / kernel_h as a kernel sees it
        # ---- init_particle_pos, fluid :142-151 ----
        x_num, z_num = F(ws[0] / self.diameter), F(ws[2] / self.diameter)
        xz_num = x_num * z_num
        i = np.arange(self.N, dtype=np.int32).astype(F)
        x = i - x_num * np.floor(i / x_num)                                        # i % x_num (float modulo, sign of the divisor)
        zz = np.floor(i / x_num)
        z = zz - z_num * np.floor(zz / z_num)
        y = (i / xz_num).astype(np.int32).astype(F)                                # int(): truncation
        start = np.array(self.start, dtype=F)
        self.pos = np.stack([x, y, z], axis=1) * F(self.radius) * F(2) + start
        # ---- init_particle_pos, walls :155-195 ----
        d = F(self.diameter)
        xr, zr = x_cnt - 1, z_cnt - 1
        wp = np.zeros((self.Nb, 3), dtype=F)
        for b in range(self.Nb):
            if b < bottom:
                wp[b] = (F(b % x_cnt) * d, F(0.0), np.floor(F(b) / F(x_cnt)) * d)
            elif b < self.Nb - bottom:
                index = b - bottom
                layer = int(np.floor(F(index) / F(ring)))
                yy = d * F(layer + 1)
                index -= layer * ring
                index += 1
                xx, zc = F(0.0), F(0.0)
                if index <= xr:
                    xx, zc = F(index % xr) * d, F(0.0)
                elif index <= xr + zr:
                    xx, zc = F(xr) * d, F((index - x_cnt) % zr) * d
                elif index <= 2 * xr + zr:
                    xx, zc = F((2 * xr + zr - index) % xr + 1) * d, F(zr) * d
                elif index <= 2 * (xr + zr):
                    xx, zc = F(0.0), F((2 * (xr + zr) - index) % zr + 1) * d
                wp[b] = (xx, yy, zc)
            else:
                index = b - (self.Nb - bottom)
                wp[b] = (F(index % x_cnt) * d, F(self.box_max[1]), F(int(F(index) / F(x_cnt))) * d)
        self.wall_pos = wp
        # ---- compute_all_boundary_volume :309-320 ----
        nb = Neighbours(self, wp, wp, same=True)
        vol = np.zeros(self.Nb, dtype=F)
        for k in range(nb.kmax):
            live = k < nb.count
            j = nb.index[:, k]
            q = _norm(wp - wp[j])
            vol = np.where(live, vol + cubic_kernel(q, self.h), vol)
        with np.errstate(divide="ignore"):
            self.wall_vol = F(1.0) / vol

    def cell(self, pos):
        """get_particle_grid_index_3d :490-494"""
        return np.floor(pos / self.h).astype(np.int32)


class Neighbours:
    """for_all_neighbor / for_all_boundary_neighbor as a SET and an ORDER (ParticleSystem.py:447-469, 337-366), by testing every pair.
    `centres` walk, `others` are met; same = the two are the same species (the walker skips itself, :461 / :362)."""

    def __init__(self, sc, centres, others, same):
        g = np.array(sc.grid, dtype=np.int64)
        cc, co = sc.cell(centres).astype(np.int64), sc.cell(others).astype(np.int64)
        # what update_grid put into the lists: a particle whose 1-D index is out of range is not appended (:393-395)
        flat = co[:, 0] + co[:, 1] * (g[0] * g[2]) + co[:, 2] * g[0]                # get_particle_grid_index_1d :486-488
        listed = (flat >= 0) & (flat <= sc.C)
        # ... and a listed particle sits in the cell with that 1-D index, whatever its coordinates were (a wrapped index is a valid cell)
        fc = np.clip(flat, 0, sc.C - 1)
        lx, lz, ly = fc % g[0], (fc // g[0]) % g[2], fc // (g[0] * g[2])
        lco = np.stack([lx, ly, lz], axis=1)
        off = lco[None, :, :] - cc[:, None, :]                                       # cell of j minus cell of i
        near = np.all(np.abs(off) <= 1, axis=2)
        inside = np.all((lco >= 0) & (lco < g), axis=1)                              # :453-456 (always true for a listed cell)
        d = centres[:, None, :] - others[None, :, :]
        dist = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2])
        ok = near & inside[None, :] & listed[None, :] & ~(dist > sc.h)               # :466 / :364: skipped if norm > support_radius
        if same:
            ok &= ~np.eye(len(centres), dtype=bool)
        rank = ((off[..., 0] + 1) * 9 + (off[..., 1] + 1) * 3 + (off[..., 2] + 1)).astype(np.int64)      # dx outermost (:452)
        key = np.where(ok, rank * len(others) + np.arange(len(others))[None, :], np.iinfo(np.int64).max)
        order = np.argsort(key, axis=1, kind="stable")
        self.count = ok.sum(axis=1)
        self.kmax = int(self.count.max()) if len(centres) else 0
        self.index = order[:, :max(self.kmax, 1)]


def cubic_kernel(r, h):
    """solver_base.py:74-88"""
    q = r / h
    k = F(8) / (F(math.pi) * _ipow(h, 3))
    q2 = q * q
    q3 = q2 * q
    inner = k * (F(6) * (q3 - q2) + F(1))
    outer = F(2) * k * _ipow(F(1) - q, 3)
    return np.where((F(0) <= q) & (q <= F(0.5)), inner, np.where((F(0.5) < q) & (q <= F(1)), outer, F(0.0))).astype(F)


def cubic_kernel_derivative(r, h):
    """solver_base.py:90-103 (with the factor 6 the reference carries)"""
    r_norm = _norm(r)
    q = r_norm / h
    k = F(48) / (F(math.pi) * _ipow(h, 3))
    q2 = q * q
    with np.errstate(divide="ignore", invalid="ignore"):
        inner = ((k * F(6)) * (F(3) * q2 - F(2) * q))[..., None] * r / (h * r_norm)[..., None]
        outer = ((-k * F(6)) * _ipow(F(1) - q, 2))[..., None] * r / (h * r_norm)[..., None]
    in1 = ((F(1e-5) < q) & (q <= F(0.5)))[..., None]
    in2 = ((F(0.5) < q) & (q <= F(1)))[..., None]
    return np.where(in1, inner, np.where(in2, outer, F(0.0))).astype(F)


class Solver:
    """solver_base + wcsph_solver / dfsph_solver on a Scene; `name` picks the constants the subclass overrides"""

    def __init__(self, config):
        self.sc = sc = Scene(config)
        sol = config["solver"]
        self.name = sol["name"]
        self.N = sc.N
        self.pos = sc.pos.copy()
        self.vel = np.zeros((self.N, 3), dtype=F)
        self.dt = F(sol["delta_time"])                                             # solver_base.py:15-16
        self.kernel_h = sc.radius * 4                                              # :17 (a Python scalar)
        self.h = F(self.kernel_h)
        self.rho_0 = F(1000)                                                       # :19
        self.gravity = scene_gravity = config["scene"]["gravity"]
        self.walls = bool(sol.get("boundary_handle", True))                        # :31
        self.m = F(sc.m)
        if self.name == "wcsph":                                                   # wcsph_solver.py:17-22
            self.eps, self.c_s, self.alpha_v, self.tension_k = 0.01, 10, 0.08, 0.2
        else:                                                                      # solver_base.py:23-26
            self.eps, self.c_s, self.alpha_v, self.tension_k = 0.01, 13, 0.08, 0.5
        self.g_vec = np.array([scene_gravity * 0.0, scene_gravity * -1.0, scene_gravity * 0.0], dtype=F)
        self.warm = np.zeros(self.N, dtype=F)                                      # dfsph_solver.py:17
        self.dt2 = self.dt * self.dt                                               # :20
        self.n_div = self.n_dens = 0

    # ---- the sums of one sweep: fluid neighbours, then wall neighbours, each in walk order ---------------------------------------------
    def _fluid_sum(self, shape, term):
        acc = np.zeros((self.N,) + shape, dtype=F)
        for k in range(self.nf.kmax):
            live = k < self.nf.count
            t = term(self.nf.index[:, k])
            acc = np.where(live.reshape((-1,) + (1,) * len(shape)), acc + t, acc)
        return acc

    def _wall_sum(self, shape, term):
        acc = np.zeros((self.N,) + shape, dtype=F)
        for k in range(self.nw.kmax):
            live = k < self.nw.count
            t = term(self.nw.index[:, k])
            acc = np.where(live.reshape((-1,) + (1,) * len(shape)), acc + t, acc)
        return acc

    def prologue(self):
        """solver_base.step :136-143: the grid is rebuilt from the positions of the previous step's end"""
        self.nf = Neighbours(self.sc, self.pos, self.pos, same=True)
        self.nw = Neighbours(self.sc, self.pos, self.sc.wall_pos, same=False) if self.walls else None

    def grad_f(self, j):
        return cubic_kernel_derivative(self.pos - self.pos[j], self.h)

    def grad_w(self, b):
        return cubic_kernel_derivative(self.pos - self.sc.wall_pos[b], self.h)

    def compute_all_rho(self):
        """solver_base.py:41-72"""
        rho = np.full(self.N, F(0.001), dtype=F)
        for k in range(self.nf.kmax):
            live = k < self.nf.count
            j = self.nf.index[:, k]
            rho = np.where(live, rho + self.m * cubic_kernel(_norm(self.pos - self.pos[j]), self.h), rho)
        if self.walls:
            rb = self._wall_sum((), lambda b: self.sc.wall_vol[b] * cubic_kernel(_norm(self.pos - self.sc.wall_pos[b]), self.h))
            rho = rho + rb * self.rho_0
        self.rho = rho

    def viscosity_and_tension(self):
        """solver_base.py:170-217"""
        num = F(2 * self.alpha_v * self.kernel_h * self.c_s)
        eps_h2 = F(self.eps * self.kernel_h * self.kernel_h)

        def visc(j):
            v_ij, x_ij = self.vel - self.vel[j], self.pos - self.pos[j]
            shear = _dot(v_ij, x_ij)
            q = _norm(x_ij)
            q2 = q * q
            nu = num / (self.rho + self.rho[j])
            pi = -nu * shear / (q2 + eps_h2)
            t = (-self.m * pi)[:, None] * cubic_kernel_derivative(x_ij, self.h)
            return np.where((shear < F(0))[:, None], t, F(0.0)).astype(F)

        coef = F(-self.tension_k / self.sc.m * self.sc.m)

        def tens(j):
            q = self.pos - self.pos[j]
            return (coef * cubic_kernel(_norm(q), self.h))[:, None] * q

        self.viscosity = self._fluid_sum((3,), visc) * self.m                      # :175
        self.tension = self._fluid_sum((3,), tens) * self.m                        # :209

    # ---- wcsph_solver.py --------------------------------------------------------------------------------------------------------
    def step_wcsph(self):
        self.prologue()
        acc = np.tile(self.g_vec, (self.N, 1))                                     # reset(), solver_base.py:131-133
        self.compute_all_rho()
        rho_i = np.where(self.rho > self.rho_0, self.rho, self.rho_0)             # ti.max, wcsph_solver.py:87
        self.pressure = F(70000) * (_ipow(rho_i / self.rho_0, 7) - F(1.0))         # :88-89
        rho_2 = _ipow(self.rho, 2)

        def pgrad(j):                                                              # :102-116
            s = self.pressure / rho_2 + self.pressure[j] / _ipow(self.rho[j], 2)
            return F(0.0) - (self.m * s)[:, None] * self.grad_f(j)

        pg = self._fluid_sum((3,), pgrad)
        bacc = np.zeros((self.N, 3), dtype=F)
        if self.walls:                                                             # :80-83, :92-100
            bacc = self._wall_sum((3,), lambda b: F(0.0) - (self.sc.wall_vol[b] * self.pressure / rho_2)[:, None] * self.grad_w(b)) * self.rho_0
        self.viscosity_and_tension()
        if self.walls:                                                             # :42-47
            acc = acc + (((pg + self.viscosity) + self.tension) + bacc)
        else:
            acc = acc + ((pg + self.viscosity) + self.tension)
        self.vel = self.vel + acc * self.dt                                        # :50
        self.vel = self.vel * F(0.9998)                                            # :51
        self.pos = self.pos + self.vel * self.dt                                   # :52
        if not self.walls:                                                         # :54-63
            self._clamp(F(self.sc.diameter))
        self.acc = acc

    def _clamp(self, off):
        for a in range(3):
            lo, hi = F(self.sc.box_min[a]) + off, F(self.sc.box_max[a]) - off
            low = self.pos[:, a] <= lo
            self.pos[:, a] = np.where(low, lo, self.pos[:, a])
            self.vel[:, a] = np.where(low, self.vel[:, a] * F(-0.5), self.vel[:, a])
            high = self.pos[:, a] >= hi
            self.pos[:, a] = np.where(high, hi, self.pos[:, a])
            self.vel[:, a] = np.where(high, self.vel[:, a] * F(-0.5), self.vel[:, a])

    # ---- dfsph_solver.py --------------------------------------------------------------------------------------------------------
    def compute_all_alpha(self):
        """:32-89"""
        ssum = self._fluid_sum((3,), lambda j: self.m * self.grad_f(j))

        def sq(j):
            r = self.m * self.grad_f(j)
            return _dot(r, r)

        qsum = self._fluid_sum((), sq)
        if self.walls:
            def wterm(b):
                return (self.sc.wall_vol[b] * self.rho_0)[:, None] * self.grad_w(b)

            bsum = self._wall_sum((3,), wterm)
            bq = self._wall_sum((), lambda b: _dot(wterm(b), wterm(b)))
            den = ((_dot(ssum, ssum) + qsum) + bq) + _dot(bsum, bsum)
        else:
            den = _dot(ssum, ssum) + qsum
        with np.errstate(divide="ignore", invalid="ignore"):
            self.alpha = np.where(np.abs(den) < F(1e-6), F(0.0), self.rho / den).astype(F)

    def _correct(self, k, vel, gate):
        """the three pressure-like corrections share one shape (:302-391, :178-219): v_i -= dt * (sum_F m (k_i/rho_i + k_j/rho_j) grad W
        + rho_0 sum_B V_b k_i / rho_i grad W); gate: the divergence iteration's `> 1e-5` (:367)"""
        def fterm(j):
            s = k / self.rho + k[j] / self.rho[j]
            t = (self.m * s)[:, None] * self.grad_f(j)
            return np.where((s > F(1e-5))[:, None], t, F(0.0)).astype(F) if gate else t

        a = self._fluid_sum((3,), fterm)
        if self.walls:
            b = self._wall_sum((3,), lambda w: (self.sc.wall_vol[w] * k / self.rho)[:, None] * self.grad_w(w))
            return vel - (a + b * self.rho_0) * self.dt
        return vel - a * self.dt

    def _residual(self, vel):
        """sum_F m (v_i - v_j) . grad W  [+ rho_0 * sum_B V_b v_i . grad W]     (:280-300, :151-176)"""
        a = self._fluid_sum((), lambda j: self.m * _dot(vel - vel[j], self.grad_f(j)))
        if self.walls:
            b = self._wall_sum((), lambda w: self.sc.wall_vol[w] * _dot(vel, self.grad_w(w)))
            return a + b * self.rho_0
        return a

    def derivative_iter_all_rho(self):
        """:252-279; get_neighbour_count (ParticleSystem.py:424-445) counts the same set for_all_neighbor walks when there is no rigid body"""
        r = self._residual(self.vel)
        r = np.where(r > F(0.0), r, F(0.0))                                        # ti.max(., 0.0)
        self.rho_derivative = np.where(self.nf.count < 20, F(0.0), r).astype(F)
        pos = self.rho_derivative[self.rho_derivative > 0]
        return float(F(math.fsum(float(v) for v in pos) / len(pos))) if len(pos) else 0.0

    def correct_divergence_error(self):
        """:393-416"""
        past = 0
        iter_cnt = 0
        self.vel = self._correct(self.warm / self.dt, self.vel, gate=False)        # divergence_warm_start :314-355
        self.warm = np.zeros(self.N, dtype=F)
        avg = self.derivative_iter_all_rho()
        self.div_first = avg
        while (iter_cnt < 1 or avg > 10) and iter_cnt < 15:
            self.vel = self._correct(self.rho_derivative * self.alpha / self.dt, self.vel, gate=True)      # :302-312, 357-391
            self.warm = self.warm + self.rho_derivative * self.alpha               # sum_up_stiff :381-384
            past = avg
            avg = self.derivative_iter_all_rho()
            if abs(avg - past) < 1e-5:
                break
            iter_cnt += 1
        self.n_div, self.div_err = iter_cnt, avg

    def step_dfsph(self):
        self.prologue()                                                            # solver_base.step; reset() is a no-op here (:418-421)
        self.compute_all_rho()                                                     # initialize :423-426
        self.compute_all_alpha()
        self.correct_divergence_error()                                            # iterate :428-438
        self.viscosity_and_tension()                                               # compute_all_ext_force :91-96 (tension first)
        force_ext = (self.g_vec + self.tension) + self.viscosity
        va = self.vel + self.dt * force_ext / self.m                               # compute_all_vel_adv :98-122
        max_vel = F(max(float(v) for v in _norm(va)))
        max_dt = F(0.4 * self.sc.radius * 2) / max_vel * F(0.2)
        self.dt = F(1e-3) if max_dt > F(1e-3) else (max_dt if max_dt > F(1e-5) else F(1e-5))
        self.dt2 = _ipow(self.dt, 2)
        rho_avg, it = math.inf, 0                                                  # correct_density_error :221-233
        while it < 2 or rho_avg - 1000 > 0.1 * 1000 * 0.01:
            r = self.rho + self.dt * self._residual(va)                            # compute_all_rho_adv :124-149
            self.rho_adv = np.where(r > self.rho_0, r, self.rho_0).astype(F)
            sel = self.rho_adv[self.rho_adv != self.rho_0]
            rho_avg = float(F(math.fsum(float(v) for v in sel) / len(sel))) if len(sel) else 1000.0
            va = self._correct((self.rho_adv - self.rho_0) * self.alpha / self.dt2, va, gate=False)        # iter_all_vel_adv :178-219
            it += 1
            if it >= 200:
                raise RuntimeError("density loop does not converge")
        self.n_dens, self.dens_err = it, rho_avg - 1000
        self.pos = self.pos + self.dt * va * F(0.9999)                             # compute_all_position :235-250
        self.vel = va * F(0.9999)
        if not self.walls:
            self._clamp(F(self.sc.radius))
        self.vel_adv = va

    def step(self):
        self.step_wcsph() if self.name == "wcsph" else self.step_dfsph()
