"""iisph_solver with the reference's surface (iisph_solver.py:5-29, 340-347).  The relaxed-Jacobi pressure solve
(:85-108) runs inside the native library with the reference's thresholds; `last_stats` keeps what the reference
prints (n_dens = l, dens_err = residual; n_div = 1 if the loop left on "Iteration trend to divergence")."""
from . import _native as nat
from .fields import DeviceField
from .solver_base import solver_base


class iisph_solver(solver_base):
    _kind = "iisph"

    def __init__(self, particle_system, config, verbose=True):
        super().__init__(particle_system, config)
        self.omega = 0.5                                # iisph_solver.py:26-29
        self.max_iter_cnt = 180
        self.min_iter_cnt = 1
        self.rho_err_percent = .1
        self.verbose = verbose
        self.v_adv = DeviceField(self, nat.F_VEL_ADV)
        self.d_ii = DeviceField(self, nat.F_D_II)
        self.d_ij = DeviceField(self, nat.F_D_IJ)
        self.a_ii = DeviceField(self, nat.F_A_II)
        self.rho_adv = DeviceField(self, nat.F_RHO_ADV)
        self.p_iter = DeviceField(self, nat.F_PRESS_ITER)
        self.f_press = DeviceField(self, nat.F_PRESS_FORCE)
        self.last_stats = None

    def step(self, nsteps=1):
        self._forward_attributes()
        st = self._sim.step_iisph(nsteps)
        self.last_stats = st
        if self.verbose:
            if st.n_div:
                print("Iteration trend to divergence")                                       # :98
            print("Iter cnt: ", st.n_dens, st.dens_err)                                      # :102
        return st
