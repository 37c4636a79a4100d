"""solver_base: what the fluid solvers share at the Python level (solver_base.py:5-39,136-143).
The shared sweeps (density, viscosity, tension, kernels) are HIP code in csrc/sph_kernels.h."""
from . import _native as nat
from .fields import DeviceField, ScalarField


class solver_base:
    _kind = None   # "wcsph" | "dfsph"
    # Attributes of the reference's solver objects that a caller may edit after construction (`solver.tension_k = 1.0`).  Taichi bakes a Python
    # scalar it meets inside a kernel into that kernel when it first compiles -- the first step() -- so _BAKED attributes are forwarded to the
    # library once, at the first step(), and later edits are ignored as the reference ignores them; _LIVE attributes are read by Python-scope
    # loops at every step (dfsph_solver.py:225, :396-404) and are forwarded whenever they changed.
    _BAKED = ("viscosity_c_s", "viscosity_alpha", "viscosity_epsilon", "tension_k")
    _LIVE = ()

    def __init__(self, particle_system, config, arith=None):
        """arith: None keeps the ParticleSystem's arithmetic; "exact" / "relaxed" rebuilds its handle with that one (see ParticleSystem)."""
        solver_config = config.get("solver")
        scene_config = config.get("scene")
        self.ps = particle_system
        self._sim = particle_system._attach_solver(self._kind, arith)
        self.arith = "relaxed" if particle_system.arith == nat.ARITH_RELAXED else "exact"
        self.particle_count = particle_system.particle_num
        self.kernel_h = self.ps.particle_radius * 4          # solver_base.py:17
        self.rho_0 = 1000                                    # :19
        self.gravity = scene_config.get("gravity")           # :20
        self.v_decay_proportion = 0.5
        self.viscosity_epsilon = 0.01                        # :23-26 (wcsph overrides c_s and k)
        self.viscosity_c_s = 13
        self.viscosity_alpha = 0.08
        self.tension_k = 0.5
        self.boundary_handle = 1 if solver_config.get("boundary_handle", True) else 0   # :31
        self.fs_couple = 1 if solver_config.get("fs_couple", True) else 0               # :32
        self.artificial_friction = 0.9999                    # :37
        self.delta_time = ScalarField(lambda: self._sim.scalar(nat.S_DELTA_TIME))        # :15-16
        self._prologue_cnt = 0
        self.simulate_cnt = ScalarField(lambda: int(self._sim.scalar(nat.S_SIMULATE_CNT)) + self._prologue_cnt)   # :21
        self.rho = DeviceField(self, nat.F_RHO)               # :14
        print("\033[32m[Solver]: {}\033[0m".format(solver_config.get("name")))   # :39

    def _forward_attributes(self):
        """`solver.<attribute> = value` reaches the library here, at the head of every step() (see _BAKED / _LIVE)."""
        sent = self.__dict__.setdefault("_sent", {})
        first = not sent
        for name in (self._BAKED if first else ()) + self._LIVE:
            value = float(getattr(self, name))
            if sent.get(name) != value:
                if first and value == self._sim.param(name):       # the library starts from the reference's values
                    sent[name] = value
                    continue
                self._sim.set_param(name, value)
                sent[name] = value
        sent["_started"] = 1.0

    def compute_all_rho(self):
        """solver_base.compute_all_rho (:41-51) as a stand-alone stage (rebuilds the lists if needed)."""
        self._sim.compute_density()

    def reset(self):
        """solver_base.reset (:131-133): acc <- gravity * (0, -1, 0).  The native force sweeps start every step from that value
        themselves (wcsph_solver.py:42-47 adds to it); nothing is stored in between, so there is nothing to do here."""

    def step(self):
        """The prologue every solver's step() starts with (:136-143): count, rebuild the cell lists, reset().  The native
        `<name>_solver.step()` runs it as part of the fused step; calling it on its own leaves the grid (and the per-step
        neighbour lists) matching the current positions, e.g. before `compute_all_rho()` / `ps.get_neighbour_count()`."""
        self._prologue_cnt += 1
        self.ps.reset_grid()
        self.ps.update_grid()
        self.reset()
