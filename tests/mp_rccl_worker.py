"""Worker of tests/test_multiprocess_gpu.py::test_rccl_two_gpus: one rank of an element-sharded run with ONE GPU PER PROCESS,
halos and reductions over RCCL (ncclSend / ncclRecv pairs and ncclAllReduce in one group per iteration, or separately).
Runs only where two GPUs are visible; the first multi-GPU box that runs the suite exercises the RCCL path before bench.py does."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ["HIP_VISIBLE_DEVICES"] = os.environ.get("LOCAL_RANK", str(rank))          # before anything touches the GPU
    import torch
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    gl = dist.new_group(backend="gloo")                                                 # host-side gathers of the results
    from nekstab_amd import mesh, seed
    from nekstab_amd.capi import NekStabHip
    from nekstab_amd.sharded import ShardRank, partition_rcb
    case = mesh.load_case_npz(os.path.join(ROOT, "tests", "golden", "cylinder_case.npz"), 6)
    kw = dict(tol_helm=1e-12, tol_pres=1e-6, tol_relative=1, nproj=8, max_helm_iter=150, max_pres_iter=48)
    full = NekStabHip(case, case.meta["vert"], case.meta["nvert"], **kw)
    qx, qy = seed.add_noise(case)
    q = (qx, qy, np.zeros((case.nel, 4, 4)))
    part = partition_rcb(case, world)
    idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
    if rank == 0:
        idt = torch.tensor(list(ShardRank.new_unique_id(full.lib)), dtype=torch.uint8, device="cuda")
    dist.broadcast(idt, 0)
    sh = ShardRank(full, case, rank, world, bytes(idt.cpu().tolist()), part)
    sh.set_option("shard_graph", 0)
    ns = 10
    sh.set_nsteps(ns); full.set_nsteps(ns)
    vq, vf = sh.alloc(2)
    sh.upload(vq, *q)
    a, b = full.alloc(2)
    full.upload(a, *q)
    full.matvec(b, a, 0)
    ref = full.download(b)
    errs = []
    for fuse in [int(x) for x in os.environ.get("NSK_PREFLIGHT_FUSE", "1,0").split(",")]:
        sh.set_option("rccl_fuse", fuse)
        sh.matvec(vf, vq, 0)
        loc = sh.download_local(vf)
        gathered = [None] * world
        dist.gather_object((sh.elems, loc), gathered if rank == 0 else None, dst=0, group=gl)
        if rank == 0:
            got = [np.empty_like(r) for r in ref]
            for elems, l in gathered:
                for g, x in zip(got, l):
                    g[elems] = x
            sc = max(np.abs(ref[0]).max(), np.abs(ref[1]).max())
            errs.append(max(np.abs(g - r).max() for g, r in zip(got[:2], ref[:2])) / sc)
    # Krylov inner product across the ranks (nsk_orth: all-reduce of the coefficients on the stream)
    h, beta = sh.orth(vf, [vq])
    if rank == 0:
        fh, fb = full.orth(b, [a])
        errs.append(abs(h[0] - fh[0]) / max(abs(fh[0]), 1e-300))
        errs.append(abs(beta - fb) / fb)
        print("MPRCCL errors", errs, flush=True)
        assert max(errs) < 1e-7, errs
    dist.barrier()
    sh.close(); full.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
