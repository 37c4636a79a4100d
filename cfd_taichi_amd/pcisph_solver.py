"""pcisph_solver with the reference's surface (pcisph_solver.py:5-47, 252-259).  `delta` / `beta` are computed at
construction like pre_compute() does (:28-47); the prediction-correction loop (:49-71) runs inside the native
library with the reference's thresholds, and what the reference prints per step is kept in `last_stats`
(n_dens = iter_cnt, dens_err = rho_err_avg)."""
from . import _native as nat
from .fields import DeviceField, ScalarField
from .solver_base import solver_base


class pcisph_solver(solver_base):
    _kind = "pcisph"

    def __init__(self, particle_system, config, verbose=True):
        super().__init__(particle_system, config)
        self.rho_max_err_percent = .1                   # pcisph_solver.py:19-21
        self.min_iteration = 1
        self.max_iteration = 80
        self.verbose = verbose
        self.beta = self._sim.scalar(nat.S_PCISPH_BETA)                     # :23
        self.delta = ScalarField(lambda: self._sim.scalar(nat.S_PCISPH_DELTA))   # :24, :47
        self.press_iter = DeviceField(self, nat.F_PRESS_ITER)
        self.press_force = DeviceField(self, nat.F_PRESS_FORCE)
        self.pos_predict = DeviceField(self, nat.F_POS_PREDICT)
        self.rho_predict = DeviceField(self, nat.F_RHO_ADV)
        self.last_stats = None
        print("PCISPH parameter delta: {}, beta: {}".format(self.delta[None], self.beta))   # :38

    def step(self, nsteps=1):
        self._forward_attributes()
        st = self._sim.step_pcisph(nsteps)
        self.last_stats = st
        if self.verbose:
            print("\t\tIter cnt: {}, error: {}".format(st.n_dens, st.dens_err))              # :71
        return st
