"""CPU suite: the voxeliser that stands in for trimesh `voxelized(pitch)[.fill()].points` (ParticleSystem.py:42-50).
Parity with trimesh itself is unpinned (not installable); what is checked: the documented algorithm's properties -- voxel centres
on the global lattice, round-to-nearest occupancy (up to pitch/2 outside the mesh), independence of the triangulation, enclosed
volume against the analytic one, and every sample point against an independent inside test."""
import os

import numpy as np
import pytest

from cfd_taichi_amd import mesh

import meshes

PITCH = 0.05
CUBE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfd_taichi_amd", "assets", "cube1.stl")


def block(points):
    k = np.round(points / PITCH).astype(int)
    assert np.allclose(points, k * PITCH, atol=1e-12)            # voxel centres are lattice points
    return k


@pytest.mark.parametrize("scale,dims", [(1.0, (17, 11, 21)), (0.5, (9, 6, 11)), (0.6, (11, 7, 13)), (3.3, (54, 34, 67))])
def test_cube1_general_path_gives_the_full_block(scale, dims):
    """No box shortcut any more: the 12-triangle cube goes through subdivide / round / fill.  Extents that are not a multiple of the
    pitch occupy the voxel their surface rounds to (scale 0.6: x ends at 0.48 -> voxel 10; scale 3.3: 2.64 -> voxel 53)."""
    v, f = mesh.load_mesh(CUBE)
    pts = mesh.voxelize(v * scale, f, PITCH)
    k = block(pts)
    assert len(pts) == dims[0] * dims[1] * dims[2]
    assert tuple(k.max(0) - k.min(0) + 1) == dims and tuple(k.min(0)) == (0, 0, 0)
    assert len(np.unique(k, axis=0)) == len(k)
    # C order of the voxel index: x slowest, z fastest
    assert np.array_equal(k, np.stack(np.meshgrid(*[np.arange(d) for d in dims], indexing="ij"), -1).reshape(-1, 3))


def test_result_does_not_depend_on_the_triangulation():
    a = mesh.voxelize(*meshes.box((0.8, 0.5, 1.0), subdiv=1), PITCH)
    b = mesh.voxelize(*meshes.box((0.8, 0.5, 1.0), subdiv=7), PITCH)
    v, f = mesh.load_mesh(CUBE)
    c = mesh.voxelize(v, f, PITCH)
    assert np.array_equal(a, b) and np.allclose(a, c, atol=1e-7)      # the STL stores 0.8 as f32


def test_surface_only_voxelisation():
    """solid.fill = false (ParticleSystem.py:48-49): the shell of the block."""
    v, f = mesh.load_mesh(CUBE)
    shell = block(mesh.voxelize(v, f, PITCH, fill=False))
    assert len(shell) == 17 * 11 * 21 - 15 * 9 * 19
    on_face = (shell == 0) | (shell == np.array([16, 10, 20]))
    assert on_face.any(axis=1).all()


def test_icosphere_volume_and_membership():
    r, c = 0.4, np.array([1.0, 0.8, 0.6])
    v, f = meshes.icosphere(r, level=3, centre=c)
    pts = mesh.voxelize(v, f, PITCH)
    block(pts)
    dist = np.linalg.norm(pts - c, axis=1)
    assert dist.max() <= r + 0.87 * PITCH                          # at most half a voxel diagonal outside
    # independent inside test on the whole lattice: every lattice point well inside the sphere is present
    lo, hi = np.floor((c - r) / PITCH).astype(int) - 1, np.ceil((c + r) / PITCH).astype(int) + 1
    lat = np.stack(np.meshgrid(*[np.arange(lo[a], hi[a] + 1) for a in range(3)], indexing="ij"), -1).reshape(-1, 3)
    deep = lat[np.linalg.norm(lat * PITCH - c, axis=1) <= r - 0.87 * PITCH]
    have = {tuple(k) for k in np.round(pts / PITCH).astype(int)}
    assert all(tuple(k) in have for k in deep)
    vol = len(pts) * PITCH ** 3
    # every voxel whose cube touches the surface is occupied: the solid grows by 0.5-0.87 pitch, (1 + 0.033 / 0.4)^3 ~ 1.27
    assert 1.0 < vol / (4.0 / 3.0 * np.pi * r ** 3) < 1.35


def test_tilted_box_membership():
    R = meshes.rot_zyx(0.5, -0.3, 0.8)
    size = np.array([0.6, 0.35, 0.45])
    off = np.array([1.0, 1.0, 1.0])
    v, f = meshes.box(size, subdiv=1, rotation=R, offset=off)
    pts = mesh.voxelize(v, f, PITCH)
    local = (pts - off) @ R                                        # back into the box frame
    slack = 0.87 * PITCH
    assert np.all(local >= -slack) and np.all(local <= size + slack)
    lo, hi = np.floor(v.min(0) / PITCH).astype(int) - 1, np.ceil(v.max(0) / PITCH).astype(int) + 1
    lat = np.stack(np.meshgrid(*[np.arange(lo[a], hi[a] + 1) for a in range(3)], indexing="ij"), -1).reshape(-1, 3)
    ll = (lat * PITCH - off) @ R
    deep = lat[np.all(ll >= slack, axis=1) & np.all(ll <= size - slack, axis=1)]
    have = {tuple(k) for k in np.round(pts / PITCH).astype(int)}
    assert len(deep) > 100 and all(tuple(k) in have for k in deep)
    assert 1.0 < len(pts) * PITCH ** 3 / size.prod() < 1.7          # thin body: (0.66 x 0.41 x 0.51) / (0.6 x 0.35 x 0.45) ~ 1.46


def test_rigid_from_config_relative_asset_path(tmp_path):
    """Shipped configs name the mesh relative to the package (no absolute paths); the reference's spelling ./obj/cube1.stl resolves too."""
    from cfd_taichi_amd import scenes
    cfg = scenes.get("experiment1")
    assert not os.path.isabs(cfg["solid"]["mesh"])
    rg = mesh.rigid_from_config(cfg)
    assert rg["points"].shape == (1001, 3) and rg["vertices"].shape == (8, 3)
    cfg["solid"]["mesh"] = "./obj/cube1.stl"
    assert mesh.rigid_from_config(cfg)["points"].shape == (1001, 3)
    cfg["solid"]["fill"] = False
    assert len(mesh.rigid_from_config(cfg)["points"]) == 11 * 7 * 13 - 9 * 5 * 11


REF_OBJ = "/root/reference/obj"


@pytest.mark.skipif(not os.path.isdir(REF_OBJ), reason="the reference's mesh assets are only mounted in the build container")
@pytest.mark.parametrize("name,nv,nf,scale", [("ball.STL", 2850, 5696, 1.0), ("spot.obj", 2930, 5856, 1.0), ("cube2.STL", 8, 12, 0.5)])
def test_reference_assets_load_and_voxelise(name, nv, nf, scale):
    """The bodies the reference ships next to cube1 (binary STL, Wavefront OBJ with quads/normals): loader and voxeliser on the real
    files, read in place.  Filled voxel volume against the mesh volume (divergence theorem): the voxel layer reaches up to pitch/2
    outside the surface, so the ratio is above 1 and below (1 + pitch / size)^3-ish; the surface voxels are a subset of the filled set."""
    v, f = mesh.load_mesh(os.path.join(REF_OBJ, name))
    assert v.shape == (nv, 3) and f.shape == (nf, 3) and f.min() == 0 and f.max() == nv - 1
    v = v * scale
    tri = v[f]
    volume = abs(np.einsum("ij,ij->i", tri[:, 0], np.cross(tri[:, 1], tri[:, 2])).sum()) / 6.0
    pitch = 0.05
    filled = mesh.voxelize(v, f, pitch, fill=True)
    shell = mesh.voxelize(v, f, pitch, fill=False)
    ratio = len(filled) * pitch ** 3 / volume
    assert 1.0 < ratio < 1.5, ratio
    assert len(shell) < len(filled)
    as_set = lambda p: set(map(tuple, np.round(p / pitch).astype(int)))
    assert as_set(shell) <= as_set(filled)
    assert (filled >= v.min(0) - pitch / 2 - 1e-9).all() and (filled <= v.max(0) + pitch / 2 + 1e-9).all()
    assert np.allclose(filled / pitch, np.round(filled / pitch))          # centres on the global lattice {k * pitch}
