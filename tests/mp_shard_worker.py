"""Worker of tests/test_multiprocess_gpu.py: one rank of an element-sharded matvec whose ranks are separate PROCESSES
sharing one GPU; halos and reductions travel through torch.distributed (gloo) on the host."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _gather_fields(dist, sh, loc, ref_like, rank, world):
    gathered = [None] * world
    dist.gather_object((sh.elems, loc), gathered if rank == 0 else None, dst=0)
    if rank != 0:
        return None
    got = [np.empty_like(r) for r in ref_like]
    for elems, l in gathered:
        for g, x in zip(got, l):
            g[elems] = x
    return got


def main_r3(kind):
    """Round-3 features across processes: 'adjoint' (singular pressure: `ortho` over all ranks), 'proj' (pressure projection
    space in shards, two consecutive maps), 'nonlinear' (closed cavity: nonlinear map, set_baseflow with the CFL maximum over
    the ranks, linearised map about the new base flow)."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from nekstab_amd import mesh, seed
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import LocalParent, ShardRank, attach_host_transport, partition_rcb
    golden = os.path.join(ROOT, "tests", "golden")
    errs = []
    overlap = kind.endswith("-overlap")     # halo of the boundary workgroups on a second stream while the interior workgroups run
    kind = kind.replace("-overlap", "")
    local = kind.endswith("-local")         # rank-local set-up: this rank's sub-mesh only, global facts through dist (LocalParent.finish_dist)
    kind = kind.replace("-local", "")
    if kind in ("adjoint", "proj"):
        adj = kind == "adjoint"
        case = mesh.load_case_npz(os.path.join(golden, "cylinder_case.npz"), 8 if adj else 6, adjoint=adj)
        full = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=8,
                          max_helm_iter=150, max_pres_iter=48)
        qx, qy = seed.add_noise(case)
        q = (qx, qy, np.zeros((case.nel, case.lx1 - 2, case.lx1 - 2)))
        part = partition_rcb(case, world)
        if local:
            lp = LocalParent(case, part, rank, tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=8, max_helm_iter=150, max_pres_iter=48)
            lp.finish_dist(dist)
            assert lp.nel < 0.8 * case.nel and lp.nsteps == full.nsteps and abs(lp.dt - full.dt) < 1e-15
            sh = ShardRank(lp, case, rank, world, None, part)
        else:
            sh = ShardRank(full, case, rank, world, None, part)
        tr = attach_host_transport(sh, dist)
        if overlap:
            sh.set_option("halo_overlap", 1)
        ns = 5 if adj else 12
        sh.set_nsteps(ns); full.set_nsteps(ns)
        vq, vf = sh.alloc(2)
        sh.upload(vq, *q)
        a, b = full.alloc(2)
        if rank == 0:
            full.upload(a, *q)
        for rep in range(1 if adj else 2):
            sh.matvec(vf, vq, 1 if adj else 0)
            if rank == 0:
                full.matvec(b, a, 1 if adj else 0)
                ref = full.download(b)
            else:
                ref = [np.empty(case.x.shape), np.empty(case.x.shape), np.empty(q[2].shape)]
            got = _gather_fields(dist, sh, sh.download_local(vf), ref, rank, world)
            if rank == 0:
                sc = max(np.abs(ref[0]).max(), np.abs(ref[1]).max())
                errs.append(max(np.abs(g - r).max() for g, r in zip(got[:2], ref[:2])) / sc)
                errs.append(abs(sh.stats()["pres_iters"] - full.stats()["pres_iters"]) * 1e-12)     # same iteration counts
                full.copy(a, b)
            sh.copy(vq, vf)
    elif kind == "box3d":
        # hexahedra across processes: three velocity components per halo message, the Schwarz layers of face / edge / corner
        # neighbours on another rank, the closed box's singular pressure operator; rank-local set-up when asked for
        from nekstab_amd import mesh3d
        ubf = lambda x, y, z: np.stack([np.sin(np.pi * x) * np.cos(np.pi * y) * 0.5 + 0.2, -np.cos(np.pi * x) * np.sin(np.pi * y) * 0.5, 0.1 * np.sin(np.pi * z) + 0.0 * x])
        case = mesh3d.box_case_3d(6, 4, 3, 6, lengths=(3.0, 1.5, 1.0), re=40.0, endtime=0.05, ub_func=ubf, warp=0.05)      # all walls: no outflow
        case.ub = case.ub * case.mask
        kw3 = dict(tol_helm=1e-12, tol_pres=1e-7, tol_relative=1, nproj=4, max_helm_iter=200, max_pres_iter=96)
        full = NekStabHip(case, case.meta["vert"], case.meta["nvert"], **kw3)
        x, y, z = case.x, case.y, case.z
        q = (np.sin(1.3 * x + z) * np.cos(2.0 * y) * case.mask, np.cos(0.7 * x + 0.2) * np.sin(3.0 * y - z) * case.mask,
             np.sin(x + y) * np.cos(2.0 * z) * case.mask, np.zeros((case.nel, 4, 4, 4)))
        part = partition_rcb(case, world)
        if local:
            lp = LocalParent(case, part, rank, **kw3)
            lp.finish_dist(dist)
            assert lp.nsteps == full.nsteps and abs(lp.dt - full.dt) < 1e-15
            sh = ShardRank(lp, case, rank, world, None, part)
        else:
            sh = ShardRank(full, case, rank, world, None, part)
        tr = attach_host_transport(sh, dist)
        if overlap:
            sh.set_option("halo_overlap", 1)
        sh.set_nsteps(4); full.set_nsteps(4)
        vq, vf = sh.alloc(2); a, b = full.alloc(2)
        sh.upload3(vq, *q)
        sh.matvec(vf, vq, 0)
        if rank == 0:
            full.upload3(a, *q); full.matvec(b, a, 0); ref = full.download3(b)
        else:
            ref = [np.empty(case.x.shape) for _ in range(3)] + [np.empty(q[3].shape)]
        got = _gather_fields(dist, sh, sh.download3_local(vf), ref, rank, world)
        if rank == 0:
            sc = max(np.abs(ref[k]).max() for k in range(3))
            errs.append(max(np.abs(g - r).max() for g, r in zip(got[:3], ref[:3])) / sc)
            errs.append(abs(sh.stats()["pres_iters"] - full.stats()["pres_iters"]) * 1e-10)
    else:
        from tests.test_sharded_r3_gpu import _cavity
        c2, p2 = _cavity()
        full = NekStabHip(c2, c2.meta["vert"], c2.meta["nvert"], tol_helm=1e-12, tol_pres=1e-5, tol_relative=1, nproj=0, max_helm_iter=150, max_pres_iter=48)
        bump = 1e-2 * np.sin(np.pi * c2.x) * np.sin(np.pi * c2.y / 1.2) * c2.mask
        q = (c2.ub[0] + bump, c2.ub[1] - 0.5 * bump, p2)
        sh = ShardRank(full, c2, rank, world, None, partition_rcb(c2, world))
        tr = attach_host_transport(sh, dist)
        sh.set_nsteps(6); full.set_nsteps(6)
        vq, vf = sh.alloc(2); a, b = full.alloc(2)
        sh.upload(vq, *q)
        sh.nonlinear_map(vf, vq)
        if rank == 0:
            full.upload(a, *q); full.nonlinear_map(b, a); ref = full.download(b)
        else:
            ref = [np.empty(c2.x.shape), np.empty(c2.x.shape), np.empty(p2.shape)]
        got = _gather_fields(dist, sh, sh.download_local(vf), ref, rank, world)
        sh.set_baseflow(vf)
        if rank == 0:
            errs.append(max(np.abs(g - r).max() for g, r in zip(got[:2], ref[:2])) / np.abs(ref[0]).max())
            full.set_baseflow(b)
            errs.append(abs(full.dt - sh.dt) + abs(full.nsteps - sh.nsteps))
        sh.set_nsteps(4); full.set_nsteps(4)
        pert = (bump, 2.0 * bump, np.zeros(p2.shape))
        sh.upload(vq, *pert)
        sh.matvec(vf, vq, 0)
        if rank == 0:
            full.upload(a, *pert); full.matvec(b, a, 0); ref = full.download(b)
        got = _gather_fields(dist, sh, sh.download_local(vf), ref, rank, world)
        if rank == 0:
            errs.append(max(np.abs(g - r).max() for g, r in zip(got[:2], ref[:2])) / np.abs(ref[0]).max())
    ok = True
    if rank == 0:
        print("MPSHARD3 %s world %d: errors %s exchanges %d allreduces %d" % (kind, world, " ".join("%.2e" % e for e in errs), tr.n_exchange, tr.n_allreduce), flush=True)
        ok = max(errs) < 1e-8 and tr.n_exchange > 0
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.broadcast(flag, 0)
    sh.close(); full.close()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 1)


def main_allred_ulp():
    """One rank receives ONE all-reduce result one ulp off (sharded.HostTransport test hook): what a collective with a rank-dependent
    reduction order would deliver.  Every rank must get NSK_ECOMM (-6) from the map -- not silently diverging convergence flags --,
    and a clean map afterwards must run (the verification found nothing: the hook fires once)."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from nekstab_amd import mesh, seed
    from nekstab_amd.capi import NekStabHip, NskError
    from nekstab_amd.sharded import ShardRank, attach_host_transport, partition_rcb
    case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 6)
    full = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-10, tol_pres=1e-4, tol_relative=1, max_helm_iter=120, max_pres_iter=48)
    sh = ShardRank(full, case, rank, world, None, partition_rcb(case, world))
    tr = attach_host_transport(sh, dist)
    sh.set_nsteps(3)
    qx, qy = seed.add_noise(case)
    vq, vf = sh.alloc(2)
    sh.upload(vq, qx, qy, np.zeros((case.nel, 4, 4)))
    os.environ["NSK_TEST_ALLRED_ULP"] = str(tr.n_allreduce + 2)          # the third all-reduce from here: inside the first map
    code = 0
    try:
        sh.matvec(vf, vq, 0)
    except NskError as e:
        code = e.code
    os.environ.pop("NSK_TEST_ALLRED_ULP")
    sh.set_option("allred_verify", 24)
    sh.matvec(vf, vq, 0)                                                 # clean transport: verified again, accepted
    flag = torch.tensor([1.0 if code == -6 else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("MPALLRED code on every rank == NSK_ECOMM:", flag.item() == 1.0, "(this rank: %d)" % code, flush=True)
    sh.close(); full.close()
    sys.exit(0 if flag.item() == 1.0 else 1)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "allred-ulp":
        return main_allred_ulp()
    if len(sys.argv) > 2:
        return main_r3(sys.argv[2])
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from nekstab_amd import mesh
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardRank, attach_host_transport, partition_rcb
    from nekstab_amd.quadrature import gauss_legendre, gauss_lobatto_legendre, interp_matrix
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 6)
    modes = np.load(os.path.join(ROOT, "tests", "golden", "cylinder_modes.npz"))
    J = interp_matrix(gauss_lobatto_legendre(6)[0], gauss_legendre(4)[0])
    u = modes["dRe_u"].astype(np.float64)
    q = (u[0], u[1], J @ modes["dRe_p"].astype(np.float64) @ J.T)
    full = NekStabHip(case, case.meta["vert"], case.meta["nvert"], tol_helm=1e-12, tol_pres=1e-6, tol_relative=1,
                      schwarz_layers=2, max_helm_iter=120, max_pres_iter=48)
    part = partition_rcb(case, world)
    sh = ShardRank(full, case, rank, world, None, part)
    tr = attach_host_transport(sh, dist)
    sh.set_nsteps(nsteps)
    vq, vf = sh.alloc(2)
    sh.upload(vq, *q)
    n2 = sh.dot(vq, vq)                          # all-reduced inner product
    sh.matvec(vf, vq, 0)
    # update_hessenberg_matrix across processes: orthogonalise M q against q (coefficient all-reduced on the stream)
    vg = sh.alloc(1)[0]
    sh.copy(vg, vf)
    hq, bq = sh.orth(vg, [vq] if False else [])          # norm only
    sh.copy(vg, vf)
    sh.scal(vq, 1.0 / np.sqrt(n2))
    h1, b1 = sh.orth(vg, [vq])
    sh.scal(vq, np.sqrt(n2))
    orth_res = abs(sh.dot(vg, vq)) / np.sqrt(n2)
    loc = sh.download_local(vf)
    # the host eigensolver loop on sharded vectors: a 3-step Arnoldi factorisation (matvec + update_hessenberg_matrix)
    from nekstab_amd import krylov
    Qs = sh.alloc(4)
    sh.copy(Qs[0], vq)
    sh.scal(Qs[0], 1.0 / sh.norm(Qs[0]))
    Hs = np.zeros((4, 3))
    krylov.arnoldi_factorization(sh, Qs, Hs, 1, 3, 0)
    gathered = [None] * world
    dist.gather_object((sh.elems, loc), gathered if rank == 0 else None, dst=0)
    ok = True
    if rank == 0:
        full.set_nsteps(nsteps)
        a, b = full.alloc(2)
        full.upload(a, *q)
        full.matvec(b, a, 0)
        ref = full.download(b)
        got = [np.empty_like(r) for r in ref]
        for elems, l in gathered:
            for g, x in zip(got, l):
                g[elems] = x
        scale = max(np.abs(ref[0]).max(), np.abs(ref[1]).max())
        err = max(np.abs(g - r).max() for g, r in zip(got[:2], ref[:2])) / scale
        perr = np.abs(got[2] - ref[2]).max() / np.abs(ref[2]).max()
        dn = abs(n2 - full.dot(a, a)) / full.dot(a, a)
        # the same orthogonalisation on the single-rank context
        g1 = full.alloc(1)[0]
        full.copy(g1, b)
        full.scal(a, 1.0 / full.norm(a))
        h0, b0 = full.orth(g1, [a])
        do = max(abs(h1[0] - h0[0]) / abs(h0[0]), abs(b1 - b0) / b0)
        Qf = full.alloc(4)
        full.upload(Qf[0], *q)
        full.scal(Qf[0], 1.0 / full.norm(Qf[0]))
        Hf = np.zeros((4, 3))
        krylov.arnoldi_factorization(full, Qf, Hf, 1, 3, 0)
        dH = np.abs(Hs - Hf).max() / np.abs(Hf).max()
        print("MPSHARD world %d nsteps %d: velocity rel diff %.3e pressure %.3e norm %.1e orth %.1e (residual %.1e) Arnoldi H %.1e exchanges %d allreduces %d"
              % (world, nsteps, err, perr, dn, do, orth_res, dH, tr.n_exchange, tr.n_allreduce), flush=True)
        ok = err < 1e-9 and perr < 1e-5 and dn < 1e-12 and tr.n_exchange > 0 and do < 1e-9 and orth_res < 1e-12 and dH < 1e-8
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.broadcast(flag, 0)
    sh.close(); full.close()
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
