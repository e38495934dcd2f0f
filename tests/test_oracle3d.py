"""Pins for the 3-D oracle (oracle/linns3d.py): the reference holds no 3-D golden data, so the 3-D
restatement is anchored on the (golden-pinned) 2-D oracle through z-invariance on an extruded mesh,
plus the discrete identities the 3-D-only terms must satisfy on a fully deformed mesh."""
import numpy as np
import pytest

from nekstab_amd import mesh3d
from oracle.linns import LinNS2D
from oracle.linns3d import LinNS3D


def _inplane_case(n=6, nz=2):
    ubf = lambda x, y, z: np.stack([1.0 - 0.3 * y * y + 0.1 * np.sin(x), 0.2 * np.cos(x) * y, 0.0 * z])
    c = mesh3d.box_case_3d(3, 2, nz, n, lengths=(2.0, 1.0, 0.8), periodic=(False, False, True), outflow_xmax=True,
                           re=40.0, endtime=0.05, ub_func=ubf)
    bump = 0.06 * np.sin(np.pi * c.x / 2.0) * np.sin(np.pi * c.y)
    c.x = c.x + bump
    c.y = c.y - 0.5 * bump
    c.ub[:] = ubf(c.x, c.y, c.z)
    c.spng = 0.5 * np.clip(c.x - 1.5, 0.0, None) ** 2
    return c


def test_z_invariant_state_steps_like_2d():
    c = _inplane_case()
    o3 = LinNS3D(x=c.x, y=c.y, z=c.z, gid=c.gid, nglob=c.nglob, mask=c.mask, ub=c.ub, spng=c.spng, re=c.re,
                 endtime=c.endtime, has_outflow=c.has_outflow)
    nel2 = 6
    sl = (slice(0, nel2), 0)
    g2raw = c.gid[sl]
    _, inv = np.unique(g2raw.ravel(), return_inverse=True)
    gid2 = inv.reshape(g2raw.shape)
    o2 = LinNS2D(x=c.x[sl], y=c.y[sl], gid=gid2, nglob=int(gid2.max()) + 1, mask=c.mask[sl], ub=c.ub[:2, :nel2, 0],
                 spng=c.spng[sl], re=c.re, endtime=c.endtime, has_outflow=c.has_outflow)
    assert o3.nsteps == o2.nsteps and abs(o3.dt - o2.dt) < 1e-15
    assert abs(o3.bm1.sum() - 0.8 * o2.bm1.sum()) < 1e-12
    x2, y2 = c.x[sl], c.y[sl]
    u2 = np.sin(1.3 * x2) * np.cos(2.0 * y2) * c.mask[sl]
    v2 = np.cos(0.7 * x2 + 0.2) * np.sin(3.0 * y2) * c.mask[sl]
    m = c.lx1 - 2
    p2 = o2.J12 @ (0.3 * np.cos(x2) * y2) @ o2.J12.T
    for adjoint in (False, True):
        st2 = o2.new_state((u2, v2, p2))
        q3 = (mesh3d.extrude_field(u2, 2), mesh3d.extrude_field(v2, 2), np.zeros_like(c.x), mesh3d.extrude_pressure(p2, 2))
        st3 = o3.new_state(q3)
        for istep in range(1, 5):
            st2 = o2.step(st2, istep, adjoint)
            st3 = o3.step(st3, istep, adjoint)
        sc = np.abs(st2["u"]).max()
        for k in range(c.lx1):
            for layer in range(2):
                e = slice(layer * nel2, (layer + 1) * nel2)
                assert np.abs(st3["u"][0][e, k] - st2["u"]).max() < 1e-10 * sc
                assert np.abs(st3["u"][1][e, k] - st2["v"]).max() < 1e-10 * sc
        assert np.abs(st3["u"][2]).max() < 1e-11 * sc
        for k in range(m):
            assert np.abs(st3["p"][:nel2, k] - st2["p"]).max() < 1e-9 * np.abs(st2["p"]).max()


def test_discrete_identities_on_deformed_mesh():
    ubf = lambda x, y, z: np.stack([np.sin(x) * np.cos(y), -np.cos(x) * np.sin(y) + 0.2 * z, 0.3 * np.sin(y + z)])
    c = mesh3d.box_case_3d(2, 2, 2, 6, lengths=(1.0, 1.2, 0.9), re=30.0, endtime=0.02, ub_func=ubf, warp=0.08)
    o = LinNS3D(x=c.x, y=c.y, z=c.z, gid=c.gid, nglob=c.nglob, mask=c.mask, ub=c.ub, spng=c.spng, re=c.re,
                endtime=c.endtime, has_outflow=False, build_solvers=False)
    rng = np.random.default_rng(3)
    assert abs(o.bm1.sum() - 1.0 * 1.2 * 0.9) < 1e-9              # faces stay planar under the warp
    u, v = rng.standard_normal(c.x.shape), rng.standard_normal(c.x.shape)
    assert abs(np.sum(v * o.axhelm(u, 0.7, 1.3)) - np.sum(u * o.axhelm(v, 0.7, 1.3))) < 1e-10 * np.abs(u).sum()
    # D^T is the transpose of D
    w = [rng.standard_normal(c.x.shape) for _ in range(3)]
    p = rng.standard_normal((c.nel,) + (4, 4, 4))
    lhs = np.sum(p * o.opdiv(w))
    g = o.opgradt(p)
    rhs = sum(np.sum(g[k] * w[k]) for k in range(3))
    assert abs(lhs - rhs) < 1e-10 * abs(lhs)
    # the weak divergence of a linear (exactly representable) solenoidal field vanishes
    lin = [c.y + 2 * c.z, c.z - c.x, 3 * c.x + c.y]
    assert np.abs(o.opdiv(lin)).max() < 1e-11
    # convection of a linear field by a constant velocity: integral of c.grad(phi) over the box
    cv = [np.full_like(c.x, 0.5), np.full_like(c.x, -1.0), np.full_like(c.x, 2.0)]
    phi = 1.0 * c.x + 2.0 * c.y - 0.5 * c.z
    assert abs(o.convect(cv, phi).sum() - (0.5 - 2.0 - 1.0) * 1.0 * 1.2 * 0.9) < 1e-10
