"""Generated closed triangle meshes for the voxeliser / rigid-coupling tests (no asset files needed)."""
import numpy as np


def box(size, subdiv=1, rotation=None, offset=(0.0, 0.0, 0.0)):
    """Axis-aligned box [0, size] with every face split into subdiv x subdiv x 2 triangles, optionally rotated (3x3) and moved."""
    sx, sy, sz = size
    verts, faces = [], []

    def face(o, u, v):
        base = len(verts)
        for i in range(subdiv + 1):
            for j in range(subdiv + 1):
                verts.append(o + u * (i / subdiv) + v * (j / subdiv))
        for i in range(subdiv):
            for j in range(subdiv):
                a = base + i * (subdiv + 1) + j
                b, c, d = a + 1, a + subdiv + 1, a + subdiv + 2
                faces.append([a, c, b]); faces.append([b, c, d])
    X, Y, Z = np.array([sx, 0, 0.0]), np.array([0, sy, 0.0]), np.array([0, 0, sz])
    O = np.zeros(3)
    face(O, X, Y); face(Z, Y, X); face(O, Y, Z); face(X, Z, Y); face(O, Z, X); face(Y, X, Z)
    v = np.asarray(verts, dtype=np.float64)
    if rotation is not None:
        v = v @ np.asarray(rotation, dtype=np.float64).T
    return v + np.asarray(offset, dtype=np.float64), np.asarray(faces, dtype=np.int64)


def icosphere(radius, level=3, centre=(0.0, 0.0, 0.0)):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    f = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
         [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    v = [np.asarray(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(level):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        f = nf
    return np.asarray(v) * radius + np.asarray(centre, dtype=np.float64), np.asarray(f, dtype=np.int64)


def rot_zyx(az, ay, ax):
    cz, sz, cy, sy, cx, sx = np.cos(az), np.sin(az), np.cos(ay), np.sin(ay), np.cos(ax), np.sin(ax)
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]])
    ry = np.array([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]])
    rx = np.array([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]])
    return rz @ ry @ rx
