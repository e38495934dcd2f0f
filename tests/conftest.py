import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")
    config.addinivalue_line("markers", "lanes: lanes / band Arnoldi (nsk_clone, nsk_matvec_batch: NOT in the reference; out of the default suites, NSK_TEST_LANES=1 runs them)")


def pytest_collection_modifyitems(config, items):
    """Lanes and the band Arnoldi factorisation are this build's own (the reference runs one map per MPI job): their tests
    stay in the tree but out of the default CPU and GPU suites (the GPU suite must stay well inside the driver's limit)."""
    if os.environ.get("NSK_TEST_LANES") == "1":
        return
    skip = pytest.mark.skip(reason="lanes / band Arnoldi are not in the reference: set NSK_TEST_LANES=1 to run these tests")
    for it in items:
        if "lanes" in it.keywords:
            it.add_marker(skip)


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def make_oracle(case, **kw):
    from oracle.linns import LinNS2D
    return LinNS2D(x=case.x, y=case.y, gid=case.gid, nglob=case.nglob, mask=case.mask, ub=case.ub,
                   spng=case.spng, re=case.re, endtime=case.endtime, lxd=case.lxd,
                   has_outflow=case.has_outflow, **kw)


@pytest.fixture(scope="session")
def case6():
    from nekstab_amd import mesh
    return mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 6)


@pytest.fixture(scope="session")
def oracle6(case6):
    return make_oracle(case6)


_ORACLE8 = {}


def oracle8_direct():
    """The lx1 = 8 direct-case oracle with its sparse factorisations (12 s to build): shared by the tests that step it
    (tests/test_fuse2_gpu.py, test_persistent_gpu.py); returns (case, oracle)."""
    if "o" not in _ORACLE8:
        from nekstab_amd import mesh
        case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), 8)
        _ORACLE8["o"] = (case, make_oracle(case))
    return _ORACLE8["o"]


def oracle8_five_steps(q):
    """The oracle's map of five direct steps of `q` at lx1 = 8 (both tests use the same seeded state): computed once."""
    key = ("ref5", float(np.sum(q[0])), float(np.sum(q[2])))
    if key not in _ORACLE8:
        _ORACLE8[key] = oracle8_direct()[1].matvec(q, nsteps=5)
    return _ORACLE8[key]


@pytest.fixture(scope="session")
def oracle6_nosolve(case6):
    return make_oracle(case6, build_solvers=False)


@pytest.fixture(scope="session")
def hip6(case6):
    if not _gpu_available():
        pytest.skip("no GPU")
    from nekstab_amd.capi import NekStabHip
    h = NekStabHip(case6, case6.meta["vert"], case6.meta["nvert"], tol_helm=1e-13, tol_pres=1e-13,
                   tol_relative=0, schwarz_layers=2, max_helm_iter=120, max_pres_iter=48)
    yield h
    h.close()


@pytest.fixture(scope="session")
def modes():
    return np.load(os.path.join(GOLDEN, "cylinder_modes.npz"))


@pytest.fixture(scope="session")
def spectre():
    return np.load(os.path.join(GOLDEN, "cylinder_spectre.npz"))
