"""Minimal mesh loading and voxelisation for the rigid body of config 5 (replaces the reference's use of
trimesh, ParticleSystem.py:42-50, which is not installable here).

PARITY UNPINNED for this file: trimesh's `voxelized(pitch).fill().points` could not be run.  What is restated is trimesh's
documented default path -- subdivide the surface until every edge is at most pitch/2, occupy voxel round(vertex / pitch) for every
vertex, fill the enclosed holes of the dense grid (scipy.ndimage.binary_fill_holes), return the voxel centres index * pitch in C
order of the index -- so voxel centres lie on the global lattice {k * pitch} and may lie up to pitch/2 OUTSIDE the mesh.
For obj/cube1.STL (box 0.8 x 0.5 x 1.0, pitch 0.05) that is a 17 x 11 x 21 block; at scale 0.6 an 11 x 7 x 13 block (the box ends
at 0.48, which occupies voxel 10); at scale 3.3 a 54 x 34 x 67 block."""
import struct

import numpy as np


def load_mesh(path):
    """Returns (vertices (Nv,3) f64 unique in order of first appearance, faces (Nf,3) int).  Binary/ASCII STL and OBJ."""
    with open(path, "rb") as f:
        data = f.read()
    tris = None
    if path.lower().endswith(".obj"):
        verts, faces = [], []
        for line in data.decode("utf-8", "replace").splitlines():
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                verts.append([float(x) for x in t[1:4]])
            elif t[0] == "f":
                idx = [int(x.split("/")[0]) - 1 for x in t[1:]]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
        return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)
    n = struct.unpack("<I", data[80:84])[0] if len(data) >= 84 else -1
    if n >= 0 and len(data) == 84 + 50 * n:                      # binary STL
        rec = np.frombuffer(data, dtype=np.uint8, offset=84).reshape(n, 50)
        tris = rec[:, 12:48].copy().view("<f4").reshape(n, 3, 3).astype(np.float64)
    else:                                                         # ASCII STL
        pts = [[float(x) for x in line.split()[1:4]] for line in data.decode("utf-8", "replace").splitlines()
               if line.strip().startswith("vertex")]
        tris = np.asarray(pts, dtype=np.float64).reshape(-1, 3, 3)
    flat = tris.reshape(-1, 3)
    uniq, first, inv = np.unique(flat, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)                                     # keep first-appearance order
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    return uniq[order], rank[inv.reshape(-1)].reshape(-1, 3)


class Mesh:
    """What the caller needs of `ps.mesh` (a trimesh.Trimesh in the reference, ParticleSystem.py:42-44): `.vertices` (assigned by
    `update_mesh_vextics`, :298-299), `.faces`, and `.export(file_type='obj')` returning the Wavefront text main.py:196-200 writes."""

    def __init__(self, vertices, faces):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)

    def export(self, file_type="obj"):
        if file_type != "obj":
            raise ValueError("only file_type='obj' is provided (main.py:199)")
        lines = ["v %.8f %.8f %.8f" % tuple(v) for v in np.asarray(self.vertices, dtype=np.float64)]
        lines += ["f %d %d %d" % tuple(int(k) + 1 for k in t) for t in self.faces]
        return "\n".join(lines) + "\n"


def _subdivided_voxels(vertices, faces, pitch, max_iter=10):
    """Voxel indices hit by the vertices of the mesh subdivided until every edge is at most pitch/2 -- trimesh's
    `voxelize_subdivide(mesh, pitch, max_iter=10, edge_factor=2.0)`: four-way midpoint subdivision of every triangle that still has an
    edge longer than pitch/2 (`remesh.subdivide_to_size`), then `np.round(vertex / pitch)` (round-half-even) of every vertex."""
    v = np.asarray(vertices, dtype=np.float64)
    tri = v[np.asarray(faces, dtype=np.int64)]                     # (n, 3, 3): triangles as coordinates, no index bookkeeping
    max_edge = pitch / 2.0
    hits = []
    for _ in range(max_iter + 1):
        edge = np.sqrt(((tri[:, [1, 2, 0]] - tri) ** 2).sum(axis=2))
        too_long = (edge > max_edge).any(axis=1)
        done = tri[~too_long].reshape(-1, 3)
        if len(done):
            hits.append(np.unique(np.round(done / pitch).astype(np.int64), axis=0))
        if not too_long.any():
            break
        t = tri[too_long]
        a, b, c = t[:, 0], t[:, 1], t[:, 2]
        ab, bc, ca = (a + b) / 2.0, (b + c) / 2.0, (c + a) / 2.0   # `vertices[edges].mean(axis=1)`
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)])
    else:
        raise ValueError("max_iter exceeded: the mesh has edges longer than %g x 2^%d" % (max_edge, max_iter))
    return np.unique(np.concatenate(hits), axis=0)


def voxelize(vertices, faces, pitch, fill=True):
    """Voxel centres of `mesh.voxelized(pitch)[.fill()].points` (ParticleSystem.py:46-50) as trimesh documents the default path:
    surface voxels = round(subdivided surface vertices / pitch); `fill()` = method 'holes' = scipy.ndimage.binary_fill_holes of the
    dense occupancy grid (6-connected background); points in C order of the (x, y, z) voxel index, centre = index * pitch + origin.
    The occupied voxels are NOT clipped to the mesh: a vertex at 0.48 with pitch 0.05 occupies voxel 10 (centre 0.5)."""
    occ = _subdivided_voxels(vertices, faces, pitch)
    origin = occ.min(axis=0)
    shape = occ.max(axis=0) - origin + 1
    dense = np.zeros(shape, dtype=bool)
    dense[tuple((occ - origin).T)] = True
    if fill:
        from scipy import ndimage
        dense = ndimage.binary_fill_holes(dense)
    idx = np.column_stack(np.nonzero(dense))                       # C order: x slowest, z fastest
    return idx.astype(np.float64) * pitch + origin.astype(np.float64) * pitch


def voxelize_filled(vertices, faces, pitch):
    return voxelize(vertices, faces, pitch, fill=True)


def _resolve(path):
    """A mesh path from a config: as given (absolute, or relative to the working directory), else relative to this package
    (shipped configs say assets/cube1.stl), else the packaged asset of the same base name (the reference's ./obj/cube1.stl).
    File names match case-insensitively: coupling_demo.json:27 spells cube1.STL as cube1.stl."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    for cand in (path, os.path.join(here, path), os.path.join(here, "assets", os.path.basename(path))):
        d, b = os.path.split(cand)
        d = d or "."
        if os.path.exists(cand):
            return cand
        if os.path.isdir(d):
            for name in os.listdir(d):
                if name.lower() == b.lower():
                    return os.path.join(d, name)
    raise FileNotFoundError(path)


def rigid_from_config(config):
    """The `solid` block of a reference-style config (ParticleSystem.py:41-64) -> sample points and vertices in the mesh
    frame plus the placement parameters, as the native library and the oracle take them."""
    solid = config["solid"]
    vertices, faces = load_mesh(_resolve(solid["mesh"]))
    vertices = vertices * float(solid.get("scale", 1))                       # mesh.apply_scale, :43
    points = voxelize(vertices, faces, float(solid["voxel_radius"]) * 2, fill=bool(solid.get("fill", True)))   # :46-50
    return {"points": points.astype(np.float32), "vertices": vertices.astype(np.float32), "faces": faces,
            "rho_0": float(solid["rho_0"]), "pos_offset": [float(v) for v in solid["pos_offset"]],
            "attitude_offset": [float(v) for v in solid["attitude_offset"]], "active": bool(solid.get("active", False))}
