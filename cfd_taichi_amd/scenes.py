"""Synthetic dam-break scenes in the reference's config/*.json schema:
deterministic lattice, zero initial velocity, no RNG.  `python -m cfd_taichi_amd.scenes` writes
them under config/."""
import copy
import json
import os

_BASE = {
    "scene": {"box_min": [0.0, 0.0, 0.0], "box_max": None, "particle_radius": 0.025, "gravity": 9.8,
              "is_output_gif": False, "is_output_ply": False, "is_simulate": True},
    "solver": {"name": None, "delta_time": None, "iter_cnt": 1, "boundary_handle": True},
    "fluid": {"start_pos": [0.1, 0.1, 0.1], "water_size": None},
}


def _scene(name, dt, box_max, water_size, start_pos=(0.1, 0.1, 0.1), boundary_handle=True):
    c = copy.deepcopy(_BASE)
    c["scene"]["box_max"] = list(box_max)
    c["solver"]["name"] = name
    c["solver"]["delta_time"] = dt
    c["solver"]["boundary_handle"] = boundary_handle
    c["fluid"]["water_size"] = list(water_size)
    c["fluid"]["start_pos"] = list(start_pos)
    return c


SCENES = {
    # config 1: reference config/breaking_dam_30k.json geometry, solver overridden to wcsph (BASELINE.json)
    "breaking_dam_30k_wcsph": lambda: _scene("wcsph", 2.5e-4, [5.0, 3.0, 1.5], [1.0, 2.8, 1.3]),
    "breaking_dam_30k_dfsph": lambda: _scene("dfsph", 2.5e-4, [5.0, 3.0, 1.5], [1.0, 2.8, 1.3]),
    # reference config/dfsph_config_backup.json geometry (N = 5879)
    "dfsph_small": lambda: _scene("dfsph", 1e-3, [1.5, 3.0, 1.5], [0.7, 1.5, 0.7], start_pos=(0.3, 0.5, 0.3)),
    "wcsph_small": lambda: _scene("wcsph", 2.5e-4, [1.5, 3.0, 1.5], [0.7, 1.5, 0.7], start_pos=(0.3, 0.5, 0.3)),
    # a low, long box with the column in one corner: the mass runs along x within a few hundred steps (slab re-balancing)
    "dfsph_dam_x": lambda: _scene("dfsph", 1e-3, [2.0, 1.0, 0.6], [0.5, 0.6, 0.4], start_pos=(0.05, 0.05, 0.1)),
    "wcsph_dam_x": lambda: _scene("wcsph", 2.5e-4, [2.0, 1.0, 0.6], [0.5, 0.6, 0.4], start_pos=(0.05, 0.05, 0.1)),
    # a tiny column resting on the floor next to a wall: exercises wall neighbours from step 1
    "wcsph_tiny_wall": lambda: _scene("wcsph", 2.5e-4, [1.0, 1.0, 1.0], [0.4, 0.5, 0.4], start_pos=(0.05, 0.05, 0.05)),
    "dfsph_tiny_wall": lambda: _scene("dfsph", 1e-3, [1.0, 1.0, 1.0], [0.4, 0.5, 0.4], start_pos=(0.05, 0.05, 0.05)),
    "wcsph_tiny_clamp": lambda: _scene("wcsph", 2.5e-4, [1.0, 1.0, 1.0], [0.4, 0.5, 0.4], start_pos=(0.1, 0.1, 0.1), boundary_handle=False),
    "dfsph_tiny_clamp": lambda: _scene("dfsph", 1e-3, [1.0, 1.0, 1.0], [0.4, 0.5, 0.4], start_pos=(0.1, 0.1, 0.1), boundary_handle=False),
    # config 2-4
    "wcsph_250k": lambda: _scene("wcsph", 2.5e-4, [10.0, 6.0, 2.7], [2.5, 5.0, 2.5]),
    "dfsph_1m": lambda: _scene("dfsph", 1e-3, [16.0, 7.0, 5.2], [5.0, 5.0, 5.0]),
    "dfsph_10m": lambda: _scene("dfsph", 1e-3, [40.0, 15.0, 10.2], [10.0, 12.5, 10.0]),
    # the solvers the reference's own configs name.  breaking_dam_30k.json:12 says iisph (delta_time 0.00025 there);
    # coupling_demo.json:16 says pcisph
    "breaking_dam_30k_iisph": lambda: _scene("iisph", 2.5e-4, [5.0, 3.0, 1.5], [1.0, 2.8, 1.3]),
    "breaking_dam_30k_pcisph": lambda: _scene("pcisph", 1e-3, [5.0, 3.0, 1.5], [1.0, 2.8, 1.3]),
    "iisph_1m": lambda: _scene("iisph", 1e-3, [16.0, 7.0, 5.2], [5.0, 5.0, 5.0]),
    "pcisph_1m": lambda: _scene("pcisph", 1e-3, [16.0, 7.0, 5.2], [5.0, 5.0, 5.0]),
    "dfsph_tiny_wall_pcisph": lambda: _scene("pcisph", 1e-3, [1.0, 1.0, 1.0], [0.4, 0.5, 0.4], start_pos=(0.05, 0.05, 0.05)),
    "dfsph_tiny_wall_iisph": lambda: _scene("iisph", 1e-3, [1.0, 1.0, 1.0], [0.4, 0.5, 0.4], start_pos=(0.05, 0.05, 0.05)),
    # reference config/pbf_config_backup.json (geometry of dfsph_config_backup.json, solver pbf, dt 2.5e-4), a column on the floor
    # next to a wall (constraints and wall terms active from the first steps), the same with clamp walls, and the 30 k dam
    "pbf_small": lambda: _scene("pbf", 2.5e-4, [1.5, 3.0, 1.5], [0.7, 1.5, 0.7], start_pos=(0.3, 0.5, 0.3)),
    "pbf_tiny_wall": lambda: _scene("pbf", 2.5e-4, [1.0, 1.0, 1.0], [0.4, 0.5, 0.4], start_pos=(0.05, 0.05, 0.05)),
    "pbf_tiny_clamp": lambda: _scene("pbf", 2.5e-4, [1.0, 1.0, 1.0], [0.4, 0.5, 0.4], start_pos=(0.1, 0.1, 0.1), boundary_handle=False),
    "breaking_dam_30k_pbf": lambda: _scene("pbf", 2.5e-4, [5.0, 3.0, 1.5], [1.0, 2.8, 1.3]),
}


def _with_solid(cfg, mesh, scale, pos_offset, attitude_offset, rho_0, active=True):
    cfg["solver"]["fs_couple"] = True
    cfg["solid"] = {"mesh": mesh, "voxel_radius": 0.025, "rho_0": rho_0, "scale": scale, "pos_offset": list(pos_offset),
                    "attitude_offset": list(attitude_offset), "fill": True, "active": active}
    return cfg


# relative to the package directory (mesh._resolve); the reference's configs say ./obj/cube1.stl relative to its checkout
_CUBE = "assets/cube1.stl"

SCENES.update({
    # config 5 family: DFSPH + one rigid box (the geometry of the reference's obj/cube1.STL: 0.8 x 0.5 x 1.0)
    # small: the box stands next to the water column, 0.05 above the floor: coupling and wall contact from the first steps
    "dfsph_rigid_small": lambda: _with_solid(_scene("dfsph", 1e-3, [2.5, 2.0, 1.5], [0.6, 1.0, 1.2], start_pos=(0.1, 0.1, 0.15)),
                                             _CUBE, 0.5, [0.75, 0.1, 0.5], [0.0, 0.0, 0.0], 2000),
    "dfsph_rigid_tilted": lambda: _with_solid(_scene("dfsph", 1e-3, [2.5, 2.0, 1.5], [0.6, 1.0, 1.2], start_pos=(0.1, 0.1, 0.15)),
                                              _CUBE, 0.5, [0.8, 0.25, 0.45], [20.0, 0.0, 35.0], 500),
    # reference config/coupling_demo.json geometry with the solver switched to dfsph (BASELINE config 5 is its x3.3 scale-up)
    "coupling_demo_dfsph": lambda: _with_solid(_scene("dfsph", 1e-4, [5.0, 7.0, 2.5], [1.5, 2.0, 2.3]),
                                               _CUBE, 1.0, [2.5, 0.9, 0.7], [0.0, 0.0, 90.0], 5000),
    # the reference's four shipped solid configs, geometry and solver as shipped (the mesh is this package's copy of the cube1 geometry):
    # config/coupling_demo.json, dam_flush_cube.json (pcisph), experiment1_config.json (iisph), experiment2_config.json (wcsph)
    "coupling_demo": lambda: _with_solid(_scene("pcisph", 1e-4, [5.0, 7.0, 2.5], [1.5, 2.0, 2.3]),
                                         _CUBE, 1.0, [2.5, 0.9, 0.7], [0.0, 0.0, 90.0], 5000),
    "dam_flush_cube": lambda: _with_solid(_scene("pcisph", 1e-4, [5.0, 3.0, 1.5], [1.8, 2.8, 1.4]),
                                          _CUBE, 1.0, [3.0, 0.0, 0.2], [0.0, 0.0, 0.0], 2000),
    "experiment1": lambda: _with_solid(_scene("iisph", 2.5e-4, [2.5, 2.4, 1.5], [1.0, 2.0, 1.4]),
                                       _CUBE, 0.6, [1.8, 0.0, 0.7], [0.0, 0.0, 0.0], 200),
    "experiment2": lambda: _with_solid(_scene("wcsph", 2.5e-4, [2.5, 2.4, 1.5], [1.0, 2.0, 1.4]),
                                       _CUBE, 0.6, [1.7, 0.6, 0.7], [0.0, 0.0, 90.0], 1000),
    "dfsph_rigid_2m": lambda: _with_solid(_scene("dfsph", 1e-3, [16.0, 12.0, 8.0], [5.0, 6.6, 7.6]),
                                          _CUBE, 3.3, [8.25, 2.97, 2.31], [0.0, 0.0, 90.0], 5000),
    # the same scene with the body moved 0.5 m down the tank, clear of the water column (the reference's placement INTERSECTS the column: the
    # density loop runs into its cap there): a coupled solve that converges, for a throughput datum that means something
    "dfsph_rigid_2m_clear": lambda: _with_solid(_scene("dfsph", 1e-3, [16.0, 12.0, 8.0], [5.0, 6.6, 7.6]),
                                                _CUBE, 3.3, [8.75, 2.97, 2.31], [0.0, 0.0, 90.0], 5000),
})


def _as_shipped(cfg, drop_boundary_handle=False, solid1=False):
    """Key-level details of the reference's shipped files that exercise the config defaults (solver_base.py:31-32, main.py:70)."""
    if drop_boundary_handle:
        del cfg["solver"]["boundary_handle"]           # wcsph_config_backup.json has no such key: the default is True
    if solid1:
        # default.json and breaking_dam_demo.json name their body block "solid1": config.get('solid', {}) is empty, no body is built
        cfg["solver"]["fs_couple"] = True
        cfg["solid1"] = {"mesh": _CUBE, "voxel_radius": 0.025, "rho_0": 500, "scale": 1, "pos_offset": [4.7, 0.9, 0.7],
                         "attitude_offset": [0.0, 0.0, 90.0], "fill": True, "active": True}
    return cfg


SCENES.update({
    # the reference's remaining shipped files, geometry / solver / dt as shipped (dfsph_config_backup.json = dfsph_small and
    # pbf_config_backup.json = pbf_small above)
    "wcsph_config_backup": lambda: _as_shipped(_scene("wcsph", 5e-4, [1.5, 3.0, 1.5], [0.7, 1.5, 0.7], start_pos=(0.3, 0.5, 0.3)),
                                               drop_boundary_handle=True),
    "pcisph_config_backup": lambda: _scene("pcisph", 1.5e-4, [1.5, 3.0, 1.5], [0.7, 1.5, 0.7], start_pos=(0.3, 0.5, 0.3)),
    "iisph_config_backup": lambda: _scene("iisph", 1e-3, [1.5, 3.0, 1.5], [0.7, 1.5, 0.7], start_pos=(0.3, 0.5, 0.3)),
    # config/breaking_dam_demo.json (dfsph, clamp walls) and default.json (pcisph, clamp walls): "solid1" blocks, i.e. fluid only
    "breaking_dam_demo": lambda: _as_shipped(_scene("dfsph", 7e-4, [10.0, 7.0, 3.0], [2.0, 3.5, 2.8], start_pos=(0.2, 0.1, 0.1),
                                                    boundary_handle=False), solid1=True),
    "default": lambda: _as_shipped(_scene("pcisph", 1e-3, [7.0, 7.0, 2.5], [2.0, 3.6, 2.3], start_pos=(0.2, 0.1, 0.1),
                                          boundary_handle=False), solid1=True),
})


def get(name):
    return SCENES[name]()


def write_all(directory):
    os.makedirs(directory, exist_ok=True)
    for name in SCENES:
        with open(os.path.join(directory, name + ".json"), "w") as f:
            json.dump(get(name), f, indent=2)
            f.write("\n")


if __name__ == "__main__":
    write_all(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config"))
