"""wcsph_solver with the reference's surface (wcsph_solver.py:5-30).  step() enqueues the whole
WCSPH step (sort, neighbour lists, density+EOS sweep, force+integrate sweep) on the handle's HIP
stream and returns without synchronising, like the reference's Taichi launches."""
from . import _native as nat
from .fields import DeviceField
from .solver_base import solver_base


class wcsph_solver(solver_base):
    _kind = "wcsph"

    def __init__(self, particle_system, config, arith=None):
        super().__init__(particle_system, config, arith)
        self.viscosity_c_s = 10      # wcsph_solver.py:17-22
        self.tension_k = 0.2
        self.gamma = 7
        self.B = 70000
        self.pressure = DeviceField(self, nat.F_PRESSURE)

    def step(self, nsteps=1):
        self._forward_attributes()
        self._sim.step_wcsph(nsteps)
