"""ctypes binding of libsph_mi355x.so (include/sph_mi355x.h).

The library is the product: there is no CPU fallback and nothing here imports oracle/.
If the shared object is missing it is built with hipcc (cfd_taichi_amd/build.py); if that is
impossible, or no gfx950 device is visible when a simulation is created, a RuntimeError is raised.
"""
import ctypes
import os

import numpy as np

from . import build as _build

SPH_OK = 0
ARITH_EXACT, ARITH_RELAXED = 0, 1      # SphConfig.arith
SPH_E_INVALID, SPH_E_HIP, SPH_E_NO_DEVICE, SPH_E_OVERFLOW, SPH_E_STATE = -1, -2, -3, -4, -5
SOLVER_WCSPH, SOLVER_DFSPH, SOLVER_PCISPH, SOLVER_IISPH, SOLVER_PBF = 0, 1, 2, 3, 4
SOLVER_IDS = {"wcsph": SOLVER_WCSPH, "dfsph": SOLVER_DFSPH, "pcisph": SOLVER_PCISPH, "iisph": SOLVER_IISPH, "pbf": SOLVER_PBF}
SPECIES_FLUID, SPECIES_WALL, SPECIES_RIGID = 0, 1, 2
F_POS, F_VEL, F_ACC, F_RHO, F_PRESSURE, F_ALPHA, F_WARM_K, F_RHO_ADV, F_RHO_DER, F_VEL_ADV = range(10)
F_NBR_COUNT = 14
F_PRESS_ITER, F_PRESS_FORCE, F_POS_PREDICT, F_D_II, F_A_II, F_D_IJ = range(16, 22)
F_PBF_LAMBDA, F_PBF_DELTA_POS = 22, 23
F_WALL_POS, F_WALL_VOL = 32, 33
F_RIGID_POS, F_RIGID_VOL, F_RIGID_FORCE, F_RIGID_MASS, F_RIGID_VERT = 48, 49, 50, 51, 52
S_RIGID_CENTROID, S_RIGID_OMEGA, S_RIGID_VEL, S_RIGID_MASS, S_RIGID_INERTIA_INV = 10, 13, 16, 19, 20
S_DELTA_TIME, S_SIMULATE_CNT, S_PARTICLE_M, S_SUPPORT_RADIUS, S_PS_DELTA_TIME, S_GRAPH_LAUNCHES = range(6)
S_PCISPH_DELTA, S_PCISPH_BETA, S_PCISPH_MAX_INDEX, S_PCISPH_MAX_COUNT = range(6, 10)
S_ARITH_RELAXED = 30
S_VERLET_BUILDS = 31
# `solver.<attribute> = value` (include/sph_mi355x.h SPH_P_*): name of the reference's attribute -> id
SOLVER_PARAMS = {"density_threshold": 64, "min_iteration_density": 65, "min_iteration_density_divergence": 66, "max_iteration_density_divergence": 67,
                 "density_divergence_threshold": 68, "warm_start": 69, "adaptive_dt": 70, "max_dt": 71, "min_dt": 72,
                 "viscosity_c_s": 73, "viscosity_alpha": 74, "viscosity_epsilon": 75, "tension_k": 76}
VECTOR_FIELDS = {F_POS, F_VEL, F_ACC, F_VEL_ADV, F_WALL_POS, F_RIGID_POS, F_RIGID_FORCE, F_RIGID_VERT, F_PRESS_FORCE, F_POS_PREDICT, F_D_II, F_D_IJ,
                 F_PBF_DELTA_POS}

EXPORTS = [
    "sph_abi_version", "sph_set_comm_sized", "sph_create", "sph_destroy", "sph_get_sizes", "sph_last_error", "sph_upload", "sph_download",
    "sph_step_wcsph", "sph_step_dfsph", "sph_step_pcisph", "sph_step_iisph", "sph_step_pbf", "sph_build_neighbors", "sph_compute_density", "sph_compute_alpha",
    "sph_get_scalar", "sph_set_scalar", "sph_synchronize", "sph_overrides", "sph_profile_enable", "sph_profile_reset", "sph_profile_kernel_count",
    "sph_profile_kernel_name", "sph_profile_get", "sph_selftest_math", "sph_selftest_wave", "sph_tune_time",
    "sph_set_comm", "sph_rccl_unique_id", "sph_rccl_attach", "sph_rccl_selftest", "sph_get_stream", "sph_plan_slabs", "sph_replan_slabs", "sph_slab_set_overlap", "sph_slab_info", "sph_comm_stats", "sph_download_local", "sph_download_ids",
    "sph_create_rigid", "sph_rigid_step",
]


class SphConfig(ctypes.Structure):
    _fields_ = [
        ("box_min", ctypes.c_double * 3),
        ("box_max", ctypes.c_double * 3),
        ("particle_radius", ctypes.c_double),
        ("gravity", ctypes.c_double),
        ("delta_time", ctypes.c_double),
        ("start_pos", ctypes.c_double * 3),
        ("water_size", ctypes.c_double * 3),
        ("boundary_handle", ctypes.c_int32),
        ("fs_couple", ctypes.c_int32),
        ("solver", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("max_neighbors", ctypes.c_int32),
        ("max_wall_neighbors", ctypes.c_int32),
        ("max_density_iters", ctypes.c_int32),
        ("slab_rank", ctypes.c_int32),
        ("slab_count", ctypes.c_int32),
        ("slab_capacity", ctypes.c_int32),
        ("slab_rebalance_every", ctypes.c_int32),
        ("arith", ctypes.c_int32),
        ("slab_ghost_layers", ctypes.c_int32),
        ("slab_overlap", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 2),
    ]


class SphRigid(ctypes.Structure):
    _fields_ = [
        ("n_particles", ctypes.c_int32),
        ("n_vertices", ctypes.c_int32),
        ("points", ctypes.c_void_p),
        ("vertices", ctypes.c_void_p),
        ("rho_0", ctypes.c_double),
        ("pos_offset", ctypes.c_double * 3),
        ("attitude_offset", ctypes.c_double * 3),
        ("active", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class SphSizes(ctypes.Structure):
    _fields_ = [
        ("n_fluid", ctypes.c_int32),
        ("n_wall", ctypes.c_int32),
        ("n_rigid", ctypes.c_int32),
        ("grid", ctypes.c_int32 * 3),
        ("n_cells", ctypes.c_int32),
        ("max_neighbors", ctypes.c_int32),
        ("max_wall_neighbors", ctypes.c_int32),
    ]


class SphStepStats(ctypes.Structure):
    _fields_ = [
        ("n_div", ctypes.c_int32),
        ("n_dens", ctypes.c_int32),
        ("n_div_evals", ctypes.c_int32),
        ("capped", ctypes.c_int32),
        ("div_first_err", ctypes.c_float),
        ("div_err", ctypes.c_float),
        ("dens_err", ctypes.c_float),
        ("dt", ctypes.c_float),
        ("max_nbrs", ctypes.c_int32),
        ("max_wall_nbrs", ctypes.c_int32),
        ("lost", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


EXCHANGE_COUNTS_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                      ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32))
EXCHANGE_BUFFERS_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int32, ctypes.c_int32)
ALLREDUCE_STREAM_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32)
EXCHANGE_COUNTS_N_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                        ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32))


class SphComm(ctypes.Structure):
    _fields_ = [
        ("user", ctypes.c_void_p),
        ("exchange_counts", EXCHANGE_COUNTS_FN),
        ("exchange_buffers", EXCHANGE_BUFFERS_FN),
        ("allreduce", ALLREDUCE_FN),
        ("send_left", ctypes.c_void_p),
        ("send_right", ctypes.c_void_p),
        ("recv_left", ctypes.c_void_p),
        ("recv_right", ctypes.c_void_p),
        ("capacity", ctypes.c_size_t),
        ("on_host", ctypes.c_int32),
        ("stream_ordered", ctypes.c_int32),
        ("allreduce_stream", ALLREDUCE_STREAM_FN),
        ("reduce_buf", ctypes.c_void_p),
        ("exchange_counts_n", EXCHANGE_COUNTS_N_FN),
        ("reduce_capacity", ctypes.c_size_t),
    ]


_lib = None


def dev_mode():
    """Development overrides (SPH_LIB here, the SPH_* knobs inside the library) take effect only with SPH_DEV=1."""
    return os.environ.get("SPH_DEV") == "1"


def library_path():
    # SPH_LIB selects an alternative build of the same library (A/B experiments, tools); default is the in-tree build.
    # It is a development override: without SPH_DEV=1 a set SPH_LIB is refused, never silently honoured or silently dropped.
    alt = os.environ.get("SPH_LIB")
    if alt and not dev_mode():
        raise RuntimeError("SPH_LIB=%s is set but development overrides need SPH_DEV=1; the product loads %s only" % (alt, _build.LIB))
    return alt or _build.LIB


# the per-step path of include/sph_mi355x.h: what any implementation of the ABI exports (the device-specific entry points --
# slabs / RCCL, profiling, tuning, self-tests -- are bound by load() for libsph_mi355x.so only)
ABI_VERSION = 5          # include/sph_mi355x.h SPH_ABI_VERSION: checked when a library is bound
CORE_EXPORTS = [
    "sph_abi_version", "sph_create", "sph_create_rigid", "sph_destroy", "sph_get_sizes", "sph_last_error", "sph_upload", "sph_download",
    "sph_step_wcsph", "sph_step_dfsph", "sph_step_pcisph", "sph_step_iisph", "sph_step_pbf", "sph_rigid_step",
    "sph_build_neighbors", "sph_compute_density", "sph_compute_alpha", "sph_get_scalar", "sph_set_scalar", "sph_synchronize",
]


def _bind_core(lib):
    vp, ci = ctypes.c_void_p, ctypes.c_int
    try:
        lib.sph_abi_version.argtypes = []
    except AttributeError:          # a build from before the symbol existed: the very case the check is for
        raise RuntimeError("%s predates SPH_ABI_VERSION (include/sph_mi355x.h is at version %d): rebuild (python -m cfd_taichi_amd.build)"
                           % (getattr(lib, "_name", "library"), ABI_VERSION))
    lib.sph_abi_version.restype = ctypes.c_int32
    if lib.sph_abi_version() != ABI_VERSION:
        raise RuntimeError("%s implements version %d of include/sph_mi355x.h, this binding version %d: rebuild (python -m cfd_taichi_amd.build)"
                           % (getattr(lib, "_name", "library"), lib.sph_abi_version(), ABI_VERSION))
    lib.sph_create.argtypes = [ctypes.POINTER(SphConfig), ctypes.POINTER(vp)]
    lib.sph_create.restype = ci
    lib.sph_create_rigid.argtypes = [ctypes.POINTER(SphConfig), ctypes.POINTER(SphRigid), ctypes.POINTER(vp)]
    lib.sph_rigid_step.argtypes = [vp]
    lib.sph_destroy.argtypes = [vp]
    lib.sph_destroy.restype = None
    lib.sph_get_sizes.argtypes = [vp, ctypes.POINTER(SphSizes)]
    lib.sph_last_error.argtypes = [vp]
    lib.sph_last_error.restype = ctypes.c_char_p
    lib.sph_upload.argtypes = [vp, ci, ci, vp, ctypes.c_size_t]
    lib.sph_download.argtypes = [vp, ci, ci, vp, ctypes.c_size_t]
    lib.sph_step_wcsph.argtypes = [vp, ci]
    lib.sph_step_pbf.argtypes = [vp, ci]
    lib.sph_step_dfsph.argtypes = [vp, ci, ctypes.POINTER(SphStepStats)]
    lib.sph_step_pcisph.argtypes = [vp, ci, ctypes.POINTER(SphStepStats)]
    lib.sph_step_iisph.argtypes = [vp, ci, ctypes.POINTER(SphStepStats)]
    for name in ("sph_build_neighbors", "sph_compute_density", "sph_compute_alpha", "sph_synchronize"):
        getattr(lib, name).argtypes = [vp]
    lib.sph_get_scalar.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double)]
    lib.sph_set_scalar.argtypes = [vp, ci, ctypes.c_double]
    return lib


def bind_core(path):
    """Any shared library that implements the per-step path of include/sph_mi355x.h (CORE_EXPORTS), ready for Simulation(cfg, lib=...):
    a test written against the ABI runs unchanged on another implementation of it."""
    lib = ctypes.CDLL(path)
    missing = [n for n in CORE_EXPORTS if not hasattr(lib, n)]
    if missing:
        raise RuntimeError("%s does not export %s" % (path, missing))
    return _bind_core(lib)


def load(build_if_missing=True):
    """Load libsph_mi355x.so; raises RuntimeError if it cannot be found or built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path) and os.environ.get("SPH_LIB"):
        raise RuntimeError("SPH_LIB=%s does not exist" % path)
    if not os.path.exists(path):
        if not build_if_missing:
            raise RuntimeError("%s is missing; run `python -m cfd_taichi_amd.build`" % path)
        _build.build()
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise RuntimeError("cannot load %s (%s); the HIP extension is required, there is no CPU fallback" % (path, e))
    vp, ci = ctypes.c_void_p, ctypes.c_int
    _bind_core(lib)
    lib.sph_profile_reset.argtypes = [vp]
    lib.sph_overrides.argtypes = [vp]
    lib.sph_overrides.restype = ctypes.c_char_p
    lib.sph_profile_enable.argtypes = [vp, ci]
    lib.sph_profile_kernel_count.argtypes = []
    lib.sph_profile_kernel_name.argtypes = [ci]
    lib.sph_profile_kernel_name.restype = ctypes.c_char_p
    lib.sph_profile_get.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]
    lib.sph_selftest_math.argtypes = [ci, ci, vp, vp, vp, ctypes.c_size_t]
    lib.sph_selftest_wave.argtypes = [ci, ci, vp, vp, ctypes.c_size_t]
    lib.sph_tune_time.argtypes = [vp, ci, ctypes.c_uint, ci, ctypes.POINTER(ctypes.c_double)]
    lib.sph_set_comm.argtypes = [vp, ctypes.POINTER(SphComm)]
    lib.sph_set_comm_sized.argtypes = [vp, ctypes.POINTER(SphComm), ctypes.c_size_t]
    lib.sph_get_stream.argtypes = [vp, ctypes.POINTER(vp)]
    lib.sph_rccl_unique_id.argtypes = [vp]
    lib.sph_rccl_attach.argtypes = [vp, vp, ctypes.c_size_t]
    lib.sph_rccl_selftest.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.c_int32, ctypes.c_int32]
    lib.sph_plan_slabs.argtypes = [ctypes.POINTER(SphConfig), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    lib.sph_slab_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    lib.sph_slab_set_overlap.argtypes = [vp, ctypes.c_int32]
    lib.sph_comm_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), ci]
    lib.sph_replan_slabs.argtypes = [ctypes.POINTER(ctypes.c_int64), ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]
    lib.sph_download_local.argtypes = [vp, ci, vp, ctypes.c_size_t]
    lib.sph_download_ids.argtypes = [vp, vp, ctypes.c_size_t]
    _lib = lib
    return lib


class SphError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libsph_mi355x error %d: %s" % (code, message))
        self.code = code


def config_from_dict(config, solver_name=None, device=0, max_neighbors=0, max_wall_neighbors=0, max_density_iters=0,
                     slab_rank=0, slab_count=0, slab_capacity=0, slab_rebalance_every=0, arith=0, slab_ghost_layers=0, slab_overlap=0):
    """Flatten a reference-style config dict (config/*.json schema) into SphConfig."""
    scene, sol, fluid = config["scene"], config["solver"], config["fluid"]
    name = solver_name or sol["name"]
    c = SphConfig()
    c.box_min[:] = [float(v) for v in scene["box_min"]]
    c.box_max[:] = [float(v) for v in scene["box_max"]]
    c.particle_radius = float(scene["particle_radius"])
    c.gravity = float(scene["gravity"])
    c.delta_time = float(sol["delta_time"])
    c.start_pos[:] = [float(v) for v in fluid["start_pos"]]
    c.water_size[:] = [float(v) for v in fluid["water_size"]]
    c.boundary_handle = 1 if sol.get("boundary_handle", True) else 0   # solver_base.py:31
    c.fs_couple = 1 if sol.get("fs_couple", True) else 0              # solver_base.py:32
    if name not in SOLVER_IDS:
        raise NotImplementedError("solver %r: this library covers %s" % (name, sorted(SOLVER_IDS)))
    c.solver = SOLVER_IDS[name]
    c.device = int(device)
    c.max_neighbors = int(max_neighbors)
    c.max_wall_neighbors = int(max_wall_neighbors)
    c.max_density_iters = int(max_density_iters)
    c.slab_rank, c.slab_count, c.slab_capacity = int(slab_rank), int(slab_count), int(slab_capacity)
    c.slab_rebalance_every = int(slab_rebalance_every)
    c.arith = int(arith)
    c.slab_ghost_layers = int(slab_ghost_layers)
    c.slab_overlap = int(slab_overlap)
    return c


def arith_id(arith):
    """"exact" / "relaxed" / 0 / 1 / None -> SphConfig.arith"""
    if arith in (None, 0, "0", "exact", ARITH_EXACT):
        return ARITH_EXACT
    if arith in (1, "1", "relaxed"):
        return ARITH_RELAXED
    raise ValueError("arith must be 'exact' or 'relaxed', got %r" % (arith,))


def plan_slabs(cfg, slab_count):
    """Host-only: (cuts, counts) of the equal-count x-slab decomposition of the scene's initial lattice."""
    lib = load()
    c = SphConfig.from_buffer_copy(cfg)
    c.slab_count = int(slab_count)
    cuts = (ctypes.c_int32 * (slab_count + 1))()
    counts = (ctypes.c_int32 * slab_count)()
    rc = lib.sph_plan_slabs(ctypes.byref(c), cuts, counts)
    if rc != SPH_OK:
        raise SphError(rc, (lib.sph_last_error(None) or b"").decode())
    return list(cuts), list(counts)


def rccl_unique_id():
    """128 opaque bytes from ncclGetUniqueId (rank 0 calls this and broadcasts them)."""
    lib = load()
    buf = ctypes.create_string_buffer(128)
    rc = lib.sph_rccl_unique_id(buf)
    if rc != SPH_OK:
        raise SphError(rc, (lib.sph_last_error(None) or b"").decode())
    return buf.raw


def replan_slabs(column_histogram, old_cuts, ghost_layers=0):
    """Host-only: the re-balancing rule of a slab run (new cuts from a per-column particle histogram; ghost_layers > 0 weighs in the ghosts of every cut)."""
    lib = load()
    gx, nslab = len(column_histogram), len(old_cuts) - 1
    hist = (ctypes.c_int64 * gx)(*[int(v) for v in column_histogram])
    old = (ctypes.c_int32 * (nslab + 1))(*[int(v) for v in old_cuts])
    new = (ctypes.c_int32 * (nslab + 1))()
    rc = lib.sph_replan_slabs(hist, gx, nslab, old, int(ghost_layers), new)
    if rc != SPH_OK:
        raise SphError(rc, (lib.sph_last_error(None) or b"").decode())
    return list(new)


class Simulation:
    """Owns one SphHandle: device buffers of one ParticleSystem + one fluid solver."""

    def __init__(self, cfg, rigid=None, lib=None):
        """rigid: dict(points, vertices, rho_0, pos_offset, attitude_offset (degrees), active) from mesh.rigid_from_config;
        lib: another implementation of the ABI's per-step path (bind_core), default libsph_mi355x.so"""
        self._lib = lib or load()
        self.cfg = cfg
        handle = ctypes.c_void_p()
        self.n_vertices = 0
        if rigid is None:
            rc = self._lib.sph_create(ctypes.byref(cfg), ctypes.byref(handle))
        else:
            pts = np.ascontiguousarray(rigid["points"], dtype=np.float32)
            vts = np.ascontiguousarray(rigid["vertices"], dtype=np.float32)
            rg = SphRigid()
            rg.n_particles, rg.n_vertices = len(pts), len(vts)
            rg.points, rg.vertices = pts.ctypes.data, vts.ctypes.data
            rg.rho_0 = float(rigid["rho_0"])
            rg.pos_offset[:] = [float(v) for v in rigid["pos_offset"]]
            rg.attitude_offset[:] = [float(v) for v in rigid["attitude_offset"]]
            rg.active = 1 if rigid.get("active", False) else 0
            rc = self._lib.sph_create_rigid(ctypes.byref(cfg), ctypes.byref(rg), ctypes.byref(handle))
            self.n_vertices = len(vts)
        if rc != SPH_OK:
            raise SphError(rc, (self._lib.sph_last_error(None) or b"").decode())
        self._h = handle
        sizes = SphSizes()
        self._check(self._lib.sph_get_sizes(self._h, ctypes.byref(sizes)))
        self.n_fluid, self.n_wall, self.n_rigid = sizes.n_fluid, sizes.n_wall, sizes.n_rigid
        self.grid = tuple(sizes.grid)
        self.n_cells = sizes.n_cells
        self.max_neighbors, self.max_wall_neighbors = sizes.max_neighbors, sizes.max_wall_neighbors
        self.last_stats = SphStepStats()

    def _check(self, rc):
        if rc != SPH_OK:
            raise SphError(rc, (self._lib.sph_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shape(self, species, field):
        if species == SPECIES_RIGID:
            n = self.n_vertices if field == F_RIGID_VERT else self.n_rigid
            return (n, 3) if field in VECTOR_FIELDS else (n,)
        n = self.n_wall if species == SPECIES_WALL else self.n_fluid
        return (n, 3) if field in VECTOR_FIELDS else (n,)

    def download(self, field, species=SPECIES_FLUID):
        out = np.empty(self._shape(species, field), dtype=np.float32)
        self._check(self._lib.sph_download(self._h, species, field, out.ctypes.data, out.size))
        return out

    def upload(self, field, values, species=SPECIES_FLUID):
        arr = np.ascontiguousarray(values, dtype=np.float32)
        if arr.shape != self._shape(species, field):
            raise ValueError("expected shape %s, got %s" % (self._shape(species, field), arr.shape))
        self._check(self._lib.sph_upload(self._h, species, field, arr.ctypes.data, arr.size))

    def step_wcsph(self, nsteps=1):
        self._check(self._lib.sph_step_wcsph(self._h, nsteps))

    def step_dfsph(self, nsteps=1):
        self._check(self._lib.sph_step_dfsph(self._h, nsteps, ctypes.byref(self.last_stats)))
        return self.last_stats

    def step_pcisph(self, nsteps=1):
        self._check(self._lib.sph_step_pcisph(self._h, nsteps, ctypes.byref(self.last_stats)))
        return self.last_stats

    def step_iisph(self, nsteps=1):
        self._check(self._lib.sph_step_iisph(self._h, nsteps, ctypes.byref(self.last_stats)))
        return self.last_stats

    def step_pbf(self, nsteps=1):
        self._check(self._lib.sph_step_pbf(self._h, nsteps))

    def step(self, nsteps=1):
        """One call for any solver; returns the last step's SphStepStats (None for wcsph)."""
        sid = self.cfg.solver
        if sid == SOLVER_WCSPH:
            return self.step_wcsph(nsteps)
        if sid == SOLVER_PBF:
            return self.step_pbf(nsteps)
        return {SOLVER_DFSPH: self.step_dfsph, SOLVER_PCISPH: self.step_pcisph, SOLVER_IISPH: self.step_iisph}[sid](nsteps)

    def rigid_step(self):
        self._check(self._lib.sph_rigid_step(self._h))

    def rigid_scalars(self):
        g = self.scalar
        return {"centroid": [g(S_RIGID_CENTROID + k) for k in range(3)], "omega": [g(S_RIGID_OMEGA + k) for k in range(3)],
                "vel": [g(S_RIGID_VEL + k) for k in range(3)], "mass": g(S_RIGID_MASS),
                "inertia_inv": [g(S_RIGID_INERTIA_INV + k) for k in range(9)]}

    def build_neighbors(self):
        self._check(self._lib.sph_build_neighbors(self._h))

    def compute_density(self):
        self._check(self._lib.sph_compute_density(self._h))

    def compute_alpha(self):
        self._check(self._lib.sph_compute_alpha(self._h))

    def scalar(self, which):
        out = ctypes.c_double()
        self._check(self._lib.sph_get_scalar(self._h, which, ctypes.byref(out)))
        return out.value

    def set_dt(self, value):
        """solver.delta_time[None] = value"""
        self._check(self._lib.sph_set_scalar(self._h, S_DELTA_TIME, float(value)))

    def set_param(self, name, value):
        """solver.<name> = value for the attributes of SOLVER_PARAMS (dfsph_solver.py:21-29, solver_base.py:23-26, wcsph_solver.py:17-20)."""
        self._check(self._lib.sph_set_scalar(self._h, SOLVER_PARAMS[name], float(value)))

    def param(self, name):
        return self.scalar(SOLVER_PARAMS[name])

    def synchronize(self):
        self._check(self._lib.sph_synchronize(self._h))

    def overrides(self):
        """Development overrides in force on this handle, e.g. ['SPH_CELL_ORDER=morton'] (needs SPH_DEV=1 to take effect at all)."""
        if not hasattr(self._lib, "sph_overrides"):
            return []
        txt = (self._lib.sph_overrides(self._h) or b"").decode()
        out = [t for t in txt.split(";") if t]
        if os.environ.get("SPH_LIB"):
            out.append("SPH_LIB=" + os.environ["SPH_LIB"])
        return out

    # ---- multi-GPU slab handles ----
    def set_comm(self, comm, struct_size=None):
        """struct_size: what a caller built against an older, shorter SphComm would pass as sizeof(SphComm) (sph_set_comm_sized): the fields
        beyond it read as 0 / NULL whatever this structure holds there."""
        self._comm = comm            # keep the callbacks and buffers alive
        if struct_size is None:
            self._check(self._lib.sph_set_comm(self._h, ctypes.byref(comm)))
        else:
            self._check(self._lib.sph_set_comm_sized(self._h, ctypes.byref(comm), int(struct_size)))

    def rccl_attach(self, unique_id, capacity_bytes=64 << 20):
        """Collective over all slabs: the library opens its own RCCL communicator (native transport, no callbacks)."""
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self._check(self._lib.sph_rccl_attach(self._h, buf, capacity_bytes))

    def rccl_selftest(self, values, op=0):
        arr = (ctypes.c_double * len(values))(*values)
        self._check(self._lib.sph_rccl_selftest(self._h, arr, len(values), op))
        return list(arr)

    def stream_ptr(self):
        """The handle's hipStream_t as an integer (for torch.cuda.ExternalStream in stream-ordered transports)."""
        out = ctypes.c_void_p()
        self._check(self._lib.sph_get_stream(self._h, ctypes.byref(out)))
        return out.value or 0

    def slab_info(self):
        out = (ctypes.c_int32 * 8)()
        self._check(self._lib.sph_slab_info(self._h, out))
        return {"owned": out[0], "ghosts": out[1], "x_lo": out[2], "x_hi": out[3], "capacity": out[4], "recuts": out[5],
                "rebalance_every": out[6], "ghost_columns": out[7] & 15, "halo_overlapped": bool(out[7] & 16), "allreduce_hidden": bool(out[7] & 32)}

    def set_slab_overlap(self, on):
        """Between steps, on every slab alike: halo and reductions of the dfsph loops on their own streams (True) or in order (False); same bits."""
        self._check(self._lib.sph_slab_set_overlap(self._h, 1 if on else 0))

    def comm_stats(self, reset=False):
        """Transport requests since the last reset: {p2p_groups, bytes_sent, bytes_received, count_exchanges, allreduce_stream, allreduce_host, steps}."""
        out = (ctypes.c_int64 * 8)()
        self._check(self._lib.sph_comm_stats(self._h, out, 1 if reset else 0))
        keys = ("p2p_groups", "bytes_sent", "bytes_received", "count_exchanges", "allreduce_stream", "allreduce_host", "steps")
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def download_local(self, field):
        """(ids, values) of every resident particle in device order; ids < 0 are ghosts (~id)."""
        info = self.slab_info()
        n = info["owned"] + info["ghosts"]
        ids = np.empty(n, dtype=np.int32)
        self._check(self._lib.sph_download_ids(self._h, ids.ctypes.data, n))
        out = np.empty((n, 3) if field in VECTOR_FIELDS else (n,), dtype=np.float32)
        self._check(self._lib.sph_download_local(self._h, field, out.ctypes.data, out.size))
        return ids, out

    def download_owned(self, field):
        ids, vals = self.download_local(field)
        keep = ids >= 0
        return ids[keep], vals[keep]

    # ---- profiling (HIP events on the handle's stream) ----
    def tune_time(self, which, lds_bytes=0, reps=10):
        out = ctypes.c_double()
        self._check(self._lib.sph_tune_time(self._h, which, lds_bytes, reps, ctypes.byref(out)))
        return out.value

    def profile_enable(self, on=True):
        self._check(self._lib.sph_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._check(self._lib.sph_profile_reset(self._h))

    def profile(self):
        """{kernel name: (total_ms, launches)} for kernels launched while profiling was on."""
        res = {}
        for k in range(self._lib.sph_profile_kernel_count()):
            ms, n = ctypes.c_double(), ctypes.c_int64()
            self._check(self._lib.sph_profile_get(self._h, k, ctypes.byref(ms), ctypes.byref(n)))
            if n.value:
                res[self._lib.sph_profile_kernel_name(k).decode()] = (ms.value, n.value)
        return res


def selftest_math(op, a, b, device=0):
    lib = load()
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty_like(a)
    rc = lib.sph_selftest_math(device, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size)
    if rc != SPH_OK:
        raise SphError(rc, (lib.sph_last_error(None) or b"").decode())
    return out


def selftest_wave(op, values, device=0):
    """Per-lane results of one wave primitive (see sph_selftest_wave); len(values) must be a multiple of 256."""
    lib = load()
    a = np.ascontiguousarray(values, dtype=np.float64)
    out = np.empty_like(a)
    rc = lib.sph_selftest_wave(device, op, a.ctypes.data, out.ctypes.data, a.size)
    if rc != SPH_OK:
        raise SphError(rc, (lib.sph_last_error(None) or b"").decode())
    return out
