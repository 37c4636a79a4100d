"""dfsph_solver with the reference's surface (dfsph_solver.py:5-30,440-445).  The host-driven
divergence / density loops of the reference (:221-233, :393-416) run inside the native library
with the same thresholds; their per-step statistics (what the reference prints) are kept in
`last_stats` instead of being printed every step."""
from . import _native as nat
from .fields import DeviceField
from .solver_base import solver_base


class dfsph_solver(solver_base):
    _kind = "dfsph"
    _BAKED = solver_base._BAKED + ("adaptive_dt", "max_dt", "min_dt")             # inside @ti.kernel compute_all_vel_adv (dfsph_solver.py:113-117)
    _LIVE = ("min_iteration_density", "density_threshold", "min_iteration_density_divergence", "max_iteration_density_divergence",
             "density_divergence_threshold", "warm_start")                         # Python-scope loops (:225, :396-404): read at every step

    def __init__(self, particle_system, config, verbose=True, arith=None):
        super().__init__(particle_system, config, arith)
        self.min_iteration_density = 2                  # dfsph_solver.py:21-29
        self.density_threshold = 0.1
        self.min_iteration_density_divergence = 1
        self.max_iteration_density_divergence = 15
        self.density_divergence_threshold = 10
        self.warm_start = True
        self.adaptive_dt = True
        self.max_dt = 1e-3
        self.min_dt = 1e-5
        self.verbose = verbose
        self.alpha = DeviceField(self, nat.F_ALPHA)
        self.rho_adv = DeviceField(self, nat.F_RHO_ADV)
        self.rho_derivative = DeviceField(self, nat.F_RHO_DER)
        self.vel_adv = DeviceField(self, nat.F_VEL_ADV)
        self.warm_start_k = DeviceField(self, nat.F_WARM_K, writable=True)
        self.last_stats = None

    def step(self, nsteps=1):
        self._forward_attributes()
        st = self._sim.step_dfsph(nsteps)
        self.last_stats = st
        if self.verbose:   # the reference prints these every step (:233, :416)
            print("[divergence iteration] count: {}, first error {}, error {}".format(st.n_div, st.div_first_err, st.div_err))
            print("[density iteration] count: {}, error {}".format(st.n_dens, st.dens_err))
        return st

    def compute_all_alpha(self):
        self._sim.compute_alpha()
