"""Mesh pipeline on the CPU: 2 x 2 refinement, z-extrusion and box meshes keep the numbering the solver
relies on -- global ids that identify exactly the coincident nodes, masks and vertex tables consistent with them."""
import numpy as np
import pytest

from nekstab_amd import mesh, mesh3d


def _coincidence_classes(coords, gid, tol=1e-5):
    """Nodes closer than `tol` (the .re2 file holds single-precision coordinates) must share their global id; returns, for
    every pair of consecutive nodes of one id, whether the two sit at one location (False only for periodic images: the
    reference's cylinder mesh is periodic in y, period 32), and the coordinate differences."""
    from scipy.spatial import cKDTree
    pts = np.stack([c.ravel() for c in coords], axis=1)
    g = gid.ravel()
    pairs = cKDTree(pts).query_pairs(tol, output_type="ndarray")
    assert np.all(g[pairs[:, 0]] == g[pairs[:, 1]])                       # coincident nodes share their id
    o2 = np.argsort(g, kind="stable")
    p2, g2 = pts[o2], g[o2]
    same_id = g2[1:] == g2[:-1]
    d = np.abs(p2[1:][same_id] - p2[:-1][same_id])
    return np.all(d < tol, axis=1), d


def test_refine_2x2_keeps_a_consistent_numbering(case6):
    fine = mesh.refine_case_2x2(case6)
    assert fine.nel == 4 * case6.nel and fine.lx1 == case6.lx1
    ok, d = _coincidence_classes((fine.x, fine.y), fine.gid)
    assert np.all(d[~ok][:, 0] < 1e-5) and np.allclose(d[~ok][:, 1], 32.0)      # the only other joins: periodic images in y
    assert fine.gid.min() == 0 and len(np.unique(fine.gid)) == fine.nglob
    # the refined mesh covers the same domain: element areas (quadrature of 1) add up
    from nekstab_amd.quadrature import gauss_lobatto_legendre
    assert abs((fine.x.max() - fine.x.min()) - (case6.x.max() - case6.x.min())) < 1e-12
    # Dirichlet nodes stay Dirichlet: the mask vanishes on the cylinder surface (r = 0.5) in both
    for c in (case6, fine):
        r = np.hypot(c.x, c.y)
        assert np.all(c.mask[np.abs(r - 0.5) < 1e-9] == 0.0)
    # vertex table: 4 corner ids per element, shared exactly between elements that share a corner location
    v = fine.meta["vert"].reshape(fine.nel, 4)
    cx = np.stack([fine.x[:, 0, 0], fine.x[:, 0, -1], fine.x[:, -1, 0], fine.x[:, -1, -1]], axis=1)
    cy = np.stack([fine.y[:, 0, 0], fine.y[:, 0, -1], fine.y[:, -1, 0], fine.y[:, -1, -1]], axis=1)
    loc = {}
    for vid, x, y in zip(v.ravel(), cx.ravel(), cy.ravel()):
        k0 = loc.setdefault(int(vid), (float(x), float(y)))
        assert abs(k0[0] - x) < 1e-5 and min(abs(k0[1] - y), abs(abs(k0[1] - y) - 32.0)) < 1e-5      # same corner, or its periodic image
    assert len(loc) == fine.meta["nvert"]


@pytest.mark.parametrize("periodic", [True, False])
def test_extrusion_numbering(case6, periodic):
    nz = 3
    c3 = mesh3d.extrude_case(case6, nz, 0.6, periodic=periodic)
    assert c3.nel == nz * case6.nel and c3.x.shape == (c3.nel,) + (6, 6, 6)
    ok, d = _coincidence_classes((c3.x, c3.y, c3.z), c3.gid)
    # ids that join two locations: periodic images only (y: period 32 from the 2-D mesh; z: period lz when extruded periodically)
    dy, dz = d[~ok][:, 1], d[~ok][:, 2]
    assert np.all(d[~ok][:, 0] < 1e-5)
    assert np.all((dy < 1e-5) | (np.abs(dy - 32.0) < 1e-5)) and np.all((dz < 1e-5) | (np.abs(dz - 0.6) < 1e-5))
    assert (np.abs(dz - 0.6) < 1e-5).any() == periodic
    assert len(np.unique(c3.gid)) == c3.nglob
    f3 = mesh3d.extrude_field(case6.ub[0], nz)
    assert f3.shape == c3.x.shape and np.array_equal(f3[: case6.nel, 0], case6.ub[0])


def test_box_case_numbering():
    c = mesh3d.box_case_3d(3, 2, 2, 6, lengths=(1.0, 0.5, 0.5), re=100.0, endtime=0.1)
    assert c.nel == 12
    assert _coincidence_classes((c.x, c.y, c.z), c.gid)[0].all()
    assert len(np.unique(c.gid)) == c.nglob
    wall = (np.abs(c.x) < 1e-12) | (np.abs(c.x - 1.0) < 1e-12) | (np.abs(c.y) < 1e-12) | (np.abs(c.y - 0.5) < 1e-12) | (np.abs(c.z) < 1e-12) | (np.abs(c.z - 0.5) < 1e-12)
    assert np.all(c.mask[wall] == 0.0) and np.all(c.mask[~wall] == 1.0)
