"""Minimal mesh loading and solid voxelisation for the rigid body of config 5 (replaces the reference's use of
trimesh, ParticleSystem.py:42-50, which is not installable here).

PARITY UNPINNED for this file: trimesh's `voxelized(pitch).fill().points` could not be run; the convention chosen is
voxel centres on the global lattice {k * pitch}, every lattice point inside or on the closed surface (the bounding
box faces included).  For obj/cube1.STL (box 0.8 x 0.5 x 1.0, pitch 0.05) that is a 17 x 11 x 21 block."""
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


def voxelize_filled(vertices, faces, pitch):
    """Lattice points {k*pitch} inside or on the closed triangle mesh (ray casting along +x, boundary counts as inside)."""
    v = np.asarray(vertices, dtype=np.float64)
    lo = np.round(v.min(0) / pitch).astype(int)
    hi = np.round(v.max(0) / pitch).astype(int)
    ks = [np.arange(lo[a], hi[a] + 1) for a in range(3)]
    gx, gy, gz = np.meshgrid(ks[0] * pitch, ks[1] * pitch, ks[2] * pitch, indexing="ij")
    pts = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1)
    tri = v[np.asarray(faces)]
    eps = 1e-6 * max(1.0, float(np.abs(v).max()))     # STL coordinates are f32: 0.8 is stored as 0.79999995
    inside = np.zeros(len(pts), dtype=bool)
    # on-surface test (distance to any triangle plane within its edges) + parity of +x ray crossings
    crossings = np.zeros(len(pts), dtype=np.int64)
    on_surface = np.zeros(len(pts), dtype=bool)
    for a, b, c in tri:
        n = np.cross(b - a, c - a)
        nn = np.linalg.norm(n)
        if nn == 0:
            continue
        # barycentric coordinates of the projection of every point along x onto the triangle's (y,z) shadow
        d = (b[1] - a[1]) * (c[2] - a[2]) - (c[1] - a[1]) * (b[2] - a[2])
        if abs(d) > eps:
            u = ((pts[:, 1] - a[1]) * (c[2] - a[2]) - (c[1] - a[1]) * (pts[:, 2] - a[2])) / d
            w = ((b[1] - a[1]) * (pts[:, 2] - a[2]) - (pts[:, 1] - a[1]) * (b[2] - a[2])) / d
            hit = (u >= -1e-9) & (w >= -1e-9) & (u + w <= 1 + 1e-9)
            xs = a[0] + u * (b[0] - a[0]) + w * (c[0] - a[0])
            on_surface |= hit & (np.abs(xs - pts[:, 0]) <= eps)
            # half-open rule on the shadow avoids double counting shared edges
            hit_strict = (u > 1e-9) & (w > 1e-9) & (u + w < 1 - 1e-9)
            crossings += (hit_strict & (xs > pts[:, 0] + eps)).astype(np.int64)
        else:
            # triangle parallel to the x axis: points lying in its plane and inside it are on the surface
            dist = np.abs((pts - a) @ n) / nn
            u_ = np.cross(b - a, pts - a) @ n
            v_ = np.cross(c - b, pts - b) @ n
            w_ = np.cross(a - c, pts - c) @ n
            on_surface |= (dist <= eps) & (u_ >= -eps) & (v_ >= -eps) & (w_ >= -eps)
    inside = on_surface | (crossings % 2 == 1)
    # lattice points on the bounding-box shell whose rays graze edges: for closed convex shells (the cube) use the bbox test
    bb = np.all((pts >= v.min(0) - eps) & (pts <= v.max(0) + eps), axis=1)
    if _is_axis_aligned_box(v, tri, eps):
        inside = bb
    return pts[inside & bb]


def _is_axis_aligned_box(v, tri, eps):
    mn, mx = v.min(0), v.max(0)
    on_face = np.any((np.abs(v - mn) <= eps) | (np.abs(v - mx) <= eps), axis=1)
    corners = np.all((np.abs(v - mn) <= eps) | (np.abs(v - mx) <= eps), axis=1)
    return bool(on_face.all() and corners.all() and len(v) == 8)


def _resolve(path):
    import os
    if os.path.exists(path):
        return path
    d, b = os.path.split(path)
    d = d or "."
    if os.path.isdir(d):                                   # the reference's configs spell cube1.STL as cube1.stl
        for name in os.listdir(d):
            if name.lower() == b.lower():
                return os.path.join(d, name)
    packaged = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", b.lower())
    if os.path.exists(packaged):
        return packaged
    raise FileNotFoundError(path)


def rigid_from_config(config):
    """The `solid` block of a reference-style config (ParticleSystem.py:41-64) -> sample points and vertices in the mesh
    frame plus the placement parameters, as the native library and the oracle take them."""
    solid = config["solid"]
    vertices, faces = load_mesh(_resolve(solid["mesh"]))
    vertices = vertices * float(solid.get("scale", 1))                       # mesh.apply_scale, :43
    if not solid.get("fill", True):
        raise NotImplementedError("solid.fill = false (surface-only voxelisation) is not built")
    points = voxelize_filled(vertices, faces, float(solid["voxel_radius"]) * 2)   # :47
    return {"points": points.astype(np.float32), "vertices": vertices.astype(np.float32), "faces": faces,
            "rho_0": float(solid["rho_0"]), "pos_offset": [float(v) for v in solid["pos_offset"]],
            "attitude_offset": [float(v) for v in solid["attitude_offset"]], "active": bool(solid.get("active", False))}
