"""ParticleSystem with the reference's constructor and attribute surface (ParticleSystem.py:29-127,
SURVEY.md section 8b), backed by a SphHandle: the lattice, the wall particles, the wall volumes
and every per-step buffer live on the MI355X; nothing is computed in Python."""
import math

import numpy as np

from . import _native as nat
from .fields import ConstField, DeviceField, ParticleFields, ScalarField


class ParticleSystem:
    material_fluid = 0            # ParticleSystem.py:74-76
    material_solid_boundary = 1
    material_solid = 2

    def __init__(self, config, device=0, max_neighbors=0, max_wall_neighbors=0, max_density_iters=0, arith=None):
        """arith: "exact" (default: every f32 operation of the reference in its order) or "relaxed" (SphConfig.arith = SPH_ARITH_RELAXED:
        the tolerance-grade pair sweeps where the library has them -- dfsph on large scenes, wcsph with Verlet lists; north_star's 1e-5 bar);
        `config["solver"]["arith"]` is read when the argument is not given.  The reference has one arithmetic; this is the one knob the
        mirror API adds."""
        self.config = config
        self.arith = nat.arith_id(arith if arith is not None else config.get("solver", {}).get("arith"))
        self._native_opts = dict(device=device, max_neighbors=max_neighbors, max_wall_neighbors=max_wall_neighbors,
                                 max_density_iters=max_density_iters, arith=self.arith)
        self._rigid_input = None
        if config.get("solid", {}):                                      # :35-64
            from . import mesh as _mesh
            self._rigid_input = _mesh.rigid_from_config(config)
            self.rigid_pos_offset = config["solid"].get("pos_offset")
            self.rigid_rho = config["solid"].get("rho_0")
            self.voxel_radius = config["solid"].get("voxel_radius")
        scene, fluid = config["scene"], config["fluid"]
        self.particle_radius = scene["particle_radius"]                 # :80
        self.particle_diameter = self.particle_radius * 2               # :81
        self.support_radius = 4 * self.particle_radius                  # :82
        self.particle_m = 1000 * (self.particle_radius ** 3) * 8        # :83
        self.water_size = np.asarray(fluid["water_size"], dtype=np.float64)
        self.start_pos = np.asarray(fluid["start_pos"], dtype=np.float64)
        self.box_max = np.asarray(scene["box_max"], dtype=np.float64)
        self.box_min = np.asarray(scene["box_min"], dtype=np.float64)
        self.rigid_particles_num = 0
        name = config["solver"].get("name")
        self._solver_kind = name if name in nat.SOLVER_IDS else ("dfsph" if self._rigid_input else "wcsph")
        self._sim = None
        self._make_sim(self._solver_kind)
        self.particle_num = self._sim.n_fluid                            # :85
        self.boundary_particles_num = self._sim.n_wall                   # :95
        self.grid_num = np.asarray(self._sim.grid, dtype=np.int32)       # :101
        self._3d_to_1d_tran = np.asarray([1, self.grid_num[0] * self.grid_num[2], self.grid_num[0]])   # :102

        self.fluid_particles = ParticleFields(
            pos=DeviceField(self, nat.F_POS, writable=True),
            vel=DeviceField(self, nat.F_VEL, writable=True),
            acc=DeviceField(self, nat.F_ACC),
            rgb=ConstField(self.particle_num, [0.0, 0.28, 1.0]),          # :227
        )
        self.boundary_particles = ParticleFields(
            pos=DeviceField(self, nat.F_WALL_POS, nat.SPECIES_WALL),
            volume=DeviceField(self, nat.F_WALL_VOL, nat.SPECIES_WALL),
        )
        self.rgba = ConstField(self.particle_num, [0.0, 0.26, 0.68, 1.0])   # :152
        self.rgb = ConstField(self.particle_num, [0.0, 0.28, 1.0])          # :117
        has_rigid = self._rigid_input is not None
        self.rigid_particles_num = self._sim.n_rigid if has_rigid else 0
        self.exist_rigid = ScalarField(lambda: 1 if has_rigid else 0)                               # :39-40
        self.active_rigid = ScalarField(lambda: 1 if has_rigid and self._rigid_input["active"] else 0)   # :63-64
        if has_rigid:
            self.rigid_vertex_count = self._sim.n_vertices
            self.rigid_particles = ParticleFields(
                pos=DeviceField(self, nat.F_RIGID_POS, nat.SPECIES_RIGID),
                volume=DeviceField(self, nat.F_RIGID_VOL, nat.SPECIES_RIGID),
                mass=DeviceField(self, nat.F_RIGID_MASS, nat.SPECIES_RIGID),
                force=DeviceField(self, nat.F_RIGID_FORCE, nat.SPECIES_RIGID),
                rgb=ConstField(self.rigid_particles_num, [1.0, 0.0, 0.0]),                          # :295
            )
            self.rigid_vertices = DeviceField(self, nat.F_RIGID_VERT, nat.SPECIES_RIGID)
            self.rigid_centriod = ScalarField(lambda: np.asarray(self._sim.rigid_scalars()["centroid"], dtype=np.float32))
            self.mesh_faces = self._rigid_input["faces"]
            self.mesh = _mesh.Mesh(self._rigid_input["vertices"], self.mesh_faces)                  # :42-44 (placed by update_mesh_vextics)
        self.delta_time = ScalarField(lambda: self._sim.scalar(nat.S_PS_DELTA_TIME))   # :37
        print("Boundary particle count: {}k".format(self.boundary_particles_num / 1000))   # :96
        print("Fluid particle count: {}k".format(self.particle_num / 1000))                # :124-127
        print("Solid particle count: {}k".format(self.rigid_particles_num / 1000))
        print("Particle mass: {}".format(self.particle_m))
        print("Grid: {}, Grid count: {}".format(self.grid_num, int(np.prod(self.grid_num))))

    # ---- native handle management -------------------------------------------------------------
    def _make_sim(self, kind):
        cfg = nat.config_from_dict(self.config, solver_name=kind, **self._native_opts)
        self._sim = nat.Simulation(cfg, rigid=self._rigid_input)
        self._solver_kind = kind

    def _attach_solver(self, kind, arith=None):
        """Called by <name>_solver.__init__: the per-solver constants (c_s, tension_k, clamp offset;
        wcsph_solver.py:17-22 vs solver_base.py:23-26) and the arithmetic are baked into the handle."""
        if arith is not None and nat.arith_id(arith) != self.arith:
            self.arith = nat.arith_id(arith)
            self._native_opts["arith"] = self.arith
            kind, self._solver_kind = kind, None            # rebuild the handle below
        if kind != self._solver_kind:
            pos = self._sim.download(nat.F_POS)
            vel = self._sim.download(nat.F_VEL)
            self._sim.close()
            self._make_sim(kind)
            self._sim.upload(nat.F_POS, pos)
            self._sim.upload(nat.F_VEL, vel)
        return self._sim

    # ---- grid API used by solver_base.step (solver_base.py:139-141) ---------------------------
    def reset_grid(self):
        """Cell lists are rebuilt from scratch by the counting sort in update_grid; nothing to clear."""

    def update_grid(self):
        self._sim.build_neighbors()

    def get_neighbour_count(self):
        """All particles at once: (N,) counts of fluid-grid entries within h (ParticleSystem.py:424-445)."""
        self._sim.build_neighbors()
        return self._sim.download(nat.F_NBR_COUNT).astype(np.int32)

    def update_mesh_vextics(self):                        # :298-299 (sic)
        """`self.mesh.vertices = self.rigid_vertices.to_numpy()`: the mesh follows the body (main.py:196-200 exports it next);
        also returns the (Nv, 3) array."""
        self.mesh.vertices = self.rigid_vertices.to_numpy()
        return self.mesh.vertices

    def compute_boundary_particles_count(self):          # :129-137
        box = self.box_max - self.box_min
        x_cnt = int(box[0] / self.particle_diameter + 1)
        z_cnt = int(box[2] / self.particle_diameter + 1)
        ring = x_cnt * z_cnt - (x_cnt - 2) * (z_cnt - 2)
        layer = int(math.ceil((box[1] - self.particle_diameter) / self.particle_diameter))
        return layer * ring + x_cnt * z_cnt * 2
