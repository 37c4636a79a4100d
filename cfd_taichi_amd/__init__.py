"""cfd_taichi_amd -- MI355X-native SPH step behind the reference's ParticleSystem / solver API.

Host code is Python (like the reference); every per-step computation runs in hand-written
gfx950 HIP kernels reached through the C-ABI of include/sph_mi355x.h (ctypes, _native.py).

    from cfd_taichi_amd import utils, ParticleSystem, wcsph_solver, dfsph_solver
    config = utils.read_config("config/breaking_dam_30k.json")
    ps = ParticleSystem(config)
    solver = dfsph_solver(ps, config)          # same discovery rule as main.py:65-68
    solver.step()
    pos = ps.fluid_particles.pos.to_numpy()    # (N, 3) f32, original particle order
"""
from . import utils  # noqa: F401
from .ParticleSystem import ParticleSystem  # noqa: F401
from .solver_base import solver_base  # noqa: F401
from .wcsph_solver import wcsph_solver  # noqa: F401
from .dfsph_solver import dfsph_solver  # noqa: F401
from .pcisph_solver import pcisph_solver  # noqa: F401
from .iisph_solver import iisph_solver  # noqa: F401
from .pbf_solver import pbf_solver  # noqa: F401
from .rigid_solver import rigid_solver  # noqa: F401

__all__ = ["utils", "ParticleSystem", "solver_base", "wcsph_solver", "dfsph_solver", "pcisph_solver", "iisph_solver", "pbf_solver", "rigid_solver"]
