"""ctypes front-end of the CPU oracle (oracle/sph_oracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: see oracle/sph_oracle.h.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

F_POS, F_VEL, F_ACC, F_RHO, F_PRESSURE, F_ALPHA, F_WARM_K, F_RHO_ADV, F_RHO_DER = range(9)
F_VEL_ADV, F_VISCOSITY, F_TENSION, F_PGRAD, F_BACC, F_NBR_COUNT, F_FORCE_EXT = range(9, 16)
F_PRESS_ITER, F_PRESS_FORCE, F_POS_PREDICT, F_D_II, F_A_II, F_D_IJ = range(16, 22)
F_PBF_LAMBDA, F_PBF_DELTA_POS = 22, 23
F_P_PAST = 24
F_WALL_POS, F_WALL_VOL = 32, 33
F_RIGID_POS, F_RIGID_VOL, F_RIGID_FORCE, F_RIGID_MASS, F_RIGID_VERT = 48, 49, 50, 51, 52
_VEC_FIELDS = {F_POS, F_VEL, F_ACC, F_VEL_ADV, F_VISCOSITY, F_TENSION, F_PGRAD, F_BACC, F_FORCE_EXT, F_WALL_POS, F_RIGID_POS, F_RIGID_FORCE, F_RIGID_VERT,
               F_PRESS_FORCE, F_POS_PREDICT, F_D_II, F_D_IJ, F_PBF_DELTA_POS}
_RIGID_FIELDS = {F_RIGID_POS, F_RIGID_VOL, F_RIGID_FORCE, F_RIGID_MASS}
_WALL_FIELDS = {F_WALL_POS, F_WALL_VOL}


class OrcConfig(ctypes.Structure):
    _fields_ = [
        ("box_min", ctypes.c_double * 3),
        ("box_max", ctypes.c_double * 3),
        ("particle_radius", ctypes.c_double),
        ("gravity", ctypes.c_double),
        ("delta_time", ctypes.c_double),
        ("start_pos", ctypes.c_double * 3),
        ("water_size", ctypes.c_double * 3),
        ("boundary_handle", ctypes.c_int),
        ("fs_couple", ctypes.c_int),
        ("solver", ctypes.c_int),
        ("num_threads", ctypes.c_int),
    ]


class OrcRigid(ctypes.Structure):
    _fields_ = [
        ("n_particles", ctypes.c_int),
        ("n_vertices", ctypes.c_int),
        ("points", ctypes.c_void_p),
        ("vertices", ctypes.c_void_p),
        ("rho_0", ctypes.c_double),
        ("pos_offset", ctypes.c_double * 3),
        ("attitude_offset_deg", ctypes.c_double * 3),
        ("active", ctypes.c_int),
    ]


class OrcStepStats(ctypes.Structure):
    _fields_ = [
        ("n_div", ctypes.c_int),
        ("n_dens", ctypes.c_int),
        ("n_div_evals", ctypes.c_int),
        ("div_first_err", ctypes.c_float),
        ("div_err", ctypes.c_float),
        ("dens_err", ctypes.c_float),
        ("dt", ctypes.c_float),
    ]


def _stale():
    libs = [os.path.join(_HERE, n) for n in ("liborc_f32.so", "liborc_f64.so", "liborc_abi.so")]
    if not all(os.path.exists(p) for p in libs):
        return True
    newest = max(os.path.getmtime(p) for p in [os.path.join(_HERE, n) for n in ("sph_oracle.c", "sph_oracle.h", "sph_oracle_abi.c", "Makefile")] +
                 [os.path.join(os.path.dirname(_HERE), "include", "sph_mi355x.h")])
    return any(os.path.getmtime(p) < newest for p in libs)


def build(force=False):
    """Compile liborc_f32.so / liborc_f64.so with gcc (oracle/Makefile)."""
    if force or _stale():
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)


ABI_LIB = os.path.join(_HERE, "liborc_abi.so")     # include/sph_mi355x.h's per-step entry points on the oracle (sph_oracle_abi.c)

_libs = {}


def _lib(precision):
    if precision not in _libs:
        build()
        lib = ctypes.CDLL(os.path.join(_HERE, "liborc_%s.so" % precision))
        lib.orc_create.restype = ctypes.c_void_p
        lib.orc_create.argtypes = [ctypes.POINTER(OrcConfig)]
        lib.orc_create_rigid.restype = ctypes.c_void_p
        lib.orc_create_rigid.argtypes = [ctypes.POINTER(OrcConfig), ctypes.POINTER(OrcRigid)]
        lib.orc_rigid_step.argtypes = [ctypes.c_void_p]
        lib.orc_rigid_step.restype = None
        lib.orc_destroy.argtypes = [ctypes.c_void_p]
        lib.orc_sizes.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        lib.orc_get.restype = ctypes.c_long
        lib.orc_get.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.orc_set.restype = ctypes.c_long
        lib.orc_set.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.orc_set_scalar.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
        lib.orc_get_scalar.restype = ctypes.c_double
        lib.orc_get_scalar.argtypes = [ctypes.c_void_p, ctypes.c_int]
        for name in ("orc_build_grid", "orc_compute_rho", "orc_compute_alpha", "orc_compute_nbr_count"):
            getattr(lib, name).argtypes = [ctypes.c_void_p]
            getattr(lib, name).restype = None
        lib.orc_step_wcsph.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.orc_step_dfsph.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(OrcStepStats)]
        lib.orc_step_pcisph.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(OrcStepStats)]
        lib.orc_step_iisph.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(OrcStepStats)]
        lib.orc_step_pbf.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.orc_set_schedule.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_int]
        lib.orc_set_schedule.restype = None
        lib.orc_cubic_kernel.restype = ctypes.c_float
        lib.orc_cubic_kernel.argtypes = [ctypes.c_float, ctypes.c_float]
        lib.orc_cubic_kernel_derivative.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]
        lib.orc_tait_pressure.restype = ctypes.c_float
        lib.orc_tait_pressure.argtypes = [ctypes.c_float]
        _libs[precision] = lib
    return _libs[precision]


def config_from_dict(config, solver=None, num_threads=1):
    """Same JSON schema as the reference's config/*.json."""
    scene, sol, fluid = config["scene"], config["solver"], config["fluid"]
    c = OrcConfig()
    c.box_min[:] = [float(v) for v in scene["box_min"]]
    c.box_max[:] = [float(v) for v in scene["box_max"]]
    c.particle_radius = float(scene["particle_radius"])
    c.gravity = float(scene["gravity"])
    c.delta_time = float(sol["delta_time"])
    c.start_pos[:] = [float(v) for v in fluid["start_pos"]]
    c.water_size[:] = [float(v) for v in fluid["water_size"]]
    c.boundary_handle = 1 if sol.get("boundary_handle", True) else 0
    c.fs_couple = 1 if sol.get("fs_couple", True) else 0
    name = solver or sol["name"]
    c.solver = {"wcsph": 0, "dfsph": 1, "pcisph": 2, "iisph": 3, "pbf": 4}[name]
    c.num_threads = int(num_threads)
    return c


class Oracle:
    """One simulation instance of the CPU restatement."""

    def __init__(self, config, solver=None, num_threads=1, precision="f32", rigid=None):
        """rigid: dict(points=(Nr,3) f32, vertices=(Nv,3) f32, rho_0, pos_offset, attitude_offset (degrees), active)"""
        self._lib = _lib(precision)
        self.cfg = config_from_dict(config, solver, num_threads)
        self.Nv = 0
        if rigid is None:
            self._h = self._lib.orc_create(ctypes.byref(self.cfg))
        else:
            pts = np.ascontiguousarray(rigid["points"], dtype=np.float32)
            vts = np.ascontiguousarray(rigid["vertices"], dtype=np.float32)
            rg = OrcRigid()
            rg.n_particles, rg.n_vertices = len(pts), len(vts)
            rg.points, rg.vertices = pts.ctypes.data, vts.ctypes.data
            rg.rho_0 = float(rigid["rho_0"])
            rg.pos_offset[:] = [float(v) for v in rigid["pos_offset"]]
            rg.attitude_offset_deg[:] = [float(v) for v in rigid["attitude_offset"]]
            rg.active = 1 if rigid.get("active", False) else 0
            self._h = self._lib.orc_create_rigid(ctypes.byref(self.cfg), ctypes.byref(rg))
            if not self._h:
                raise ValueError("oracle: rigid coupling is restated for dfsph only")
            self.Nv = len(vts)
        sz = (ctypes.c_int * 7)()
        self._lib.orc_sizes(self._h, sz)
        self.N, self.Nb, self.Nr = sz[0], sz[1], sz[2]
        self.grid = (sz[3], sz[4], sz[5])
        self.C = sz[6]
        self.last_stats = OrcStepStats()

    def close(self):
        if self._h:
            self._lib.orc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _shape(self, field):
        n = self.Nb if field in _WALL_FIELDS else (self.Nr if field in _RIGID_FIELDS else (self.Nv if field == F_RIGID_VERT else self.N))
        return (n, 3) if field in _VEC_FIELDS else (n,)

    def get(self, field):
        out = np.empty(self._shape(field), dtype=np.float32)
        n = self._lib.orc_get(self._h, field, out.ctypes.data)
        assert n == out.size, (field, n, out.size)
        return out

    def set(self, field, values):
        arr = np.ascontiguousarray(values, dtype=np.float32)
        assert arr.shape == self._shape(field)
        n = self._lib.orc_set(self._h, field, arr.ctypes.data)
        assert n == arr.size

    @property
    def dt(self):
        return self._lib.orc_get_scalar(self._h, 0)

    def set_dt(self, value):
        """delta_time, delta_time_2 and ps.delta_time as a dfsph step leaves them: continue from a state produced elsewhere."""
        assert self._lib.orc_set_scalar(self._h, 0, float(value)) == 0

    def set_param(self, which, value):
        """`solver.<attribute> = value`: which = the SPH_P_* numbers of include/sph_mi355x.h (64..76)."""
        assert self._lib.orc_set_scalar(self._h, int(which), float(value)) == 0

    @property
    def particle_m(self):
        return self._lib.orc_get_scalar(self._h, 2)

    @property
    def lost(self):
        return int(self._lib.orc_get_scalar(self._h, 4))

    def rigid_step(self):
        self._lib.orc_rigid_step(self._h)

    def rigid_scalars(self):
        g = lambda k: self._lib.orc_get_scalar(self._h, k)   # noqa: E731
        return {"centroid": [g(10), g(11), g(12)], "omega": [g(13), g(14), g(15)], "vel": [g(16), g(17), g(18)], "mass": g(19),
                "inertia_inv": [g(20 + k) for k in range(9)]}

    def set_schedule(self, seed, chunk=1):
        """seed != 0: one seeded LEGAL execution of the reference's races (cell-list append order, f32 atomic means); 0 = canonical."""
        self._lib.orc_set_schedule(self._h, int(seed), int(chunk))

    def build_grid(self):
        self._lib.orc_build_grid(self._h)

    def compute_rho(self):
        self._lib.orc_compute_rho(self._h)

    def compute_alpha(self):
        self._lib.orc_compute_alpha(self._h)

    def compute_nbr_count(self):
        self._lib.orc_compute_nbr_count(self._h)

    def step_wcsph(self, nsteps=1):
        self._lib.orc_step_wcsph(self._h, nsteps)

    def step_pcisph(self, nsteps=1):
        """pcisph_solver.step; last_stats.n_dens = pressure iterations, dens_err = rho_err_avg.  Returns 1 at max_iteration."""
        rc = self._lib.orc_step_pcisph(self._h, nsteps, ctypes.byref(self.last_stats))
        assert rc >= 0, "oracle not created with solver pcisph"
        return rc

    def step_iisph(self, nsteps=1):
        """iisph_solver.step; last_stats.n_dens = l, dens_err = residual, n_div = 1 if the loop left on 'trend to divergence'."""
        rc = self._lib.orc_step_iisph(self._h, nsteps, ctypes.byref(self.last_stats))
        assert rc >= 0, "oracle not created with solver iisph"
        return rc

    @property
    def pcisph_delta(self):
        return self._lib.orc_get_scalar(self._h, 5)

    @property
    def pcisph_max_index(self):
        return int(self._lib.orc_get_scalar(self._h, 7)), int(self._lib.orc_get_scalar(self._h, 8))

    def step_pbf(self, nsteps=1):
        """pbf_solver.step under the barrier-synchronised schedule of update_all_pos (see sph_oracle.c)."""
        rc = self._lib.orc_step_pbf(self._h, nsteps)
        assert rc == 0, "oracle not created with solver pbf"

    def step_dfsph(self, nsteps=1, max_dens_iter=0):
        """Returns 1 if the (non-reference) density-iteration cap was hit."""
        return self._lib.orc_step_dfsph(self._h, nsteps, max_dens_iter, ctypes.byref(self.last_stats))


def cubic_kernel(r, h, precision="f32"):
    return float(_lib(precision).orc_cubic_kernel(r, h))


def cubic_kernel_derivative(rvec, h, precision="f32"):
    r = np.ascontiguousarray(rvec, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    _lib(precision).orc_cubic_kernel_derivative(r.ctypes.data, h, out.ctypes.data)
    return out


def tait_pressure(rho, precision="f32"):
    return float(_lib(precision).orc_tait_pressure(rho))
