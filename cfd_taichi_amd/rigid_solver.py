"""rigid_solver with the reference's surface (rigid_solver.py:4-31, 216-232): one rigid body driven by the forces the
fluid sweeps accumulate on its sample particles.  The per-particle parts (torque/force sums, rotation, wall test,
translation) are HIP kernels; the 3x3 algebra between them runs on the host inside the native library."""
import numpy as np

from . import _native as nat
from .fields import ScalarField


class rigid_solver:
    def __init__(self, particle_system, config):
        solid_config = config.get("solid")
        if not solid_config or not particle_system.exist_rigid[None]:
            raise ValueError("rigid_solver needs a config with a 'solid' block (main.py:69-71)")
        self.ps = particle_system
        self._sim = particle_system._sim
        self.gravity = config["scene"].get("gravity")
        self.rho = solid_config.get("rho_0")
        self.particle_count = self.ps.rigid_particles_num
        self.v_decay_proportion = 0.1                       # rigid_solver.py:24
        self.simulate_cnt_host = 0
        self.simulate_cnt = ScalarField(lambda: self.simulate_cnt_host)
        self.omega = ScalarField(lambda: np.asarray(self.ps._sim.rigid_scalars()["omega"], dtype=np.float32))
        self.mass = ScalarField(lambda: self.ps._sim.rigid_scalars()["mass"])

    def step(self):
        self.simulate_cnt_host += 1
        self.ps._sim.rigid_step()
