"""Worker of tests/test_multiprocess_gpu.py: one rank of an element-sharded matvec whose ranks are separate PROCESSES
sharing one GPU; halos and reductions travel through torch.distributed (gloo) on the host."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
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
