"""pbf_solver with the reference's surface (pbf_solver.py:5-24, 176-187).  The reference file is stale at the surveyed commit (its fluid
callbacks index particle fields with what ParticleSystem.for_all_neighbor now passes as particle structs) and update_all_pos races on
pos / vel; csrc/sph_pbf_kernels.h states how both are read.  No rigid coupling."""
from . import _native as nat
from .fields import DeviceField
from .solver_base import solver_base


class pbf_solver(solver_base):
    _kind = "pbf"

    def __init__(self, particle_system, config):
        super().__init__(particle_system, config)
        self.epsilon = 1.0e-6                           # pbf_solver.py:17-21
        self.k = 1e-7
        self.c = 9e-6
        self.s_corr_factor = 0.3
        self.pbf_lambda = DeviceField(self, nat.F_PBF_LAMBDA)
        self.delta_pos = DeviceField(self, nat.F_PBF_DELTA_POS)
        self.pos_predict = DeviceField(self, nat.F_POS_PREDICT)

    def step(self, nsteps=1):
        self._sim.step_pbf(nsteps)
