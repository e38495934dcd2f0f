"""Element sharding across PROCESSES: 2 and 3 ranks, one process each, all on the one GPU of the test box;
dssum / Schwarz halos and reductions go through the same pack / unpack tables as the RCCL transport, staged through the host
over torch.distributed gloo (nsk_comm_init_host).  Equal to the single-rank result to the solver tolerance."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,port", [(2, 29631), (3, 29633)])
def test_sharded_matvec_across_processes(world, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "mp_shard_worker.py"), "4"],
                         capture_output=True, text=True, timeout=900, env=env)
    line = [l for l in out.stdout.splitlines() if l.startswith("MPSHARD")]
    print(line, out.stderr[-1500:] if out.returncode else "")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert line


@pytest.mark.parametrize("kind,world,port", [("adjoint", 2, 29641), ("proj", 3, 29643), ("nonlinear", 2, 29645),
                                             ("proj-local", 3, 29647), ("adjoint-local", 2, 29649), ("proj-local-overlap", 3, 29651),
                                             ("box3d-local-overlap", 2, 29653), ("box3d", 3, 29655), ("box3d-local", 8, 29657)])
def test_round3_shard_features_across_processes(kind, world, port):
    """VERDICT r2 item 2 across processes (host-staged transport, ranks share the GPU): adjoint cylinder at lx1 = 8 (singular
    pressure, `ortho` through the all-reduce), the pressure projection space in shards over two maps, and the closed cavity's
    nonlinear map + set_baseflow (CFL maximum over the ranks) + linearised map -- each equal to the single-rank result.
    "-local": the shard comes from the RANK-LOCAL set-up (sharded.LocalParent: this rank's sub-mesh, volume / CFL maximum /
    coarse rows exchanged through torch.distributed) instead of a whole-mesh parent.  ("box3d-local", 8): EIGHT processes (9 elements per rank, rank-local set-up,
    singular operator, projection space).  "-overlap": option halo_overlap (the
    boundary workgroups' halo travels while the interior workgroups run).  "box3d": a closed hexahedral box (singular pressure
    operator, three-component halos, Schwarz layers of face / edge / corner neighbours on other ranks)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "mp_shard_worker.py"), "0", kind],
                         capture_output=True, text=True, timeout=900, env=env)
    line = [l for l in out.stdout.splitlines() if l.startswith("MPSHARD3")]
    print(line, out.stderr[-1500:] if out.returncode else "")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert line


def test_ranks_that_disagree_on_an_all_reduce_are_refused():
    """the first all-reduces of a sharded context are verified bit for bit across the ranks (exact integer
    chunk sums: k_ar_chunks / k_ar_check); one rank fed a result one ulp off => NSK_ECOMM on EVERY rank, no divergent
    device-side convergence flags; the next map on the clean transport runs."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29671", os.path.join(ROOT, "tests", "mp_shard_worker.py"), "0", "allred-ulp"],
                         capture_output=True, text=True, timeout=600, env=env)
    line = [l for l in out.stdout.splitlines() if l.startswith("MPALLRED")]
    print(line, out.stderr[-1500:] if out.returncode else "")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert line and "True" in line[0]


def test_bench_two_ranks_under_the_launcher_dry_run():
    """`bench.py --gpus 2` exactly as the driver starts it (torch.distributed.run, one supervisor per rank, each starting its
    worker), with the ranks sharing this GPU and the halos travelling over gloo (NSK_DIST_BACKEND=gloo: the protocol dry run):
    rank-local set-up, a sharded probe map, the timed sharded Arnoldi steps, ONE JSON line from rank 0 that names the same
    workload as the N = 1 record and carries the one-GPU rate of the same steps."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NSK_DIST_BACKEND="gloo", NSK_BENCH_FORCE_MODE_PROBE="1")   # (the comparison of the sharded modes runs too)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29661", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cfg3-probe"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    print({k: r[k] for k in ("value", "n_gpus", "scaling", "setup")}, r["config"]["parallelism"])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["value"] > 0 and r["scaling"] == "strong"
    assert "BASELINE configs[1]" in r["config"]["workload"] and "E=1996, lx1=8" in r["config"]["workload"]
    assert r["setup"]["rank_local"] and r["setup"]["elements_rank0"] < 0.8 * r["setup"]["elements_mesh"]
    assert r["single_gpu_same_config"]["matvecs_per_s"] > 0
    sm = r["shard_mode"]
    assert sm["picked"] in ("graph", "hostcheck", "hostcheck_overlap") and all(sm[k] > 0 for k in ("graph", "hostcheck", "hostcheck_overlap"))
    assert sm[sm["picked"]] == min(sm[k] for k in ("graph", "hostcheck", "hostcheck_overlap"))


def test_bench_eight_ranks_gloo_dry_run():
    """`python bench.py --gpus 8`: the script starts its own eight ranks; they share this GPU and every halo
    travels over gloo (protocol dry run), on the cylinder at lx1 = 6 with maps of four time steps to keep it short.  The three-mode probe runs
    (NSK_BENCH_FORCE_MODE_PROBE), the record has the shape the driver reads and names the headline workload."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NSK_DIST_BACKEND="gloo", NSK_BENCH_FORCE_MODE_PROBE="1", NSK_BENCH_PROBE_STEPS="2", NSK_BENCH_MAP_STEPS="4")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--lx1", "6", "--no-cfg3-probe"],
                         capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    print({k: r[k] for k in ("value", "n_gpus", "scaling", "setup", "shard_mode")}, r["config"]["parallelism"])
    assert r["n_gpus"] == 8 and r["steps"] == 1 and r["value"] > 0 and r["scaling"] == "strong" and r["unit"] == "matvecs/s"
    assert "BASELINE configs[1]" in r["config"]["workload"] and "E=1996, lx1=6" in r["config"]["workload"] and "x8" in r["config"]["parallelism"]
    assert r["setup"]["rank_local"] and r["setup"]["elements_rank0"] < 0.4 * r["setup"]["elements_mesh"]
    sm = r["shard_mode"]
    assert sm["picked"] in ("graph", "hostcheck", "hostcheck_overlap") and all(sm[k] > 0 for k in ("graph", "hostcheck", "hostcheck_overlap"))
    assert r["single_gpu_same_config"]["matvecs_per_s"] > 0 and r["roofline"]["peak"] == 8000.0


def test_rccl_two_gpus():
    """RCCL between two GPUs: two processes, one GPU each, the sharded map with the all-reduce inside the
    halo group and with separate calls, and nsk_orth across the ranks -- equal to the single-rank result.  Skipped on one-GPU boxes
    (every development box so far); the first node with two GPUs runs it before bench.py --gpus 2 does."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (%d visible)" % torch.cuda.device_count())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29671", os.path.join(ROOT, "tests", "mp_rccl_worker.py")], capture_output=True, text=True, timeout=900, env=env)
    line = [l for l in out.stdout.splitlines() if l.startswith("MPRCCL")]
    print(line, out.stderr[-1500:] if out.returncode else "")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert line
