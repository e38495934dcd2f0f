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
