"""Time-stepper parity through the C-ABI: nsk_matvec vs the oracle on identical
inputs (a committed reference eigenmode as the initial perturbation)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mode(o, modes, key):
    J = o.J12
    u = modes[key + "_u"].astype(np.float64)
    p = modes[key + "_p"].astype(np.float64)
    return u[0], u[1], J @ p @ J.T


def relL2(o, a, b):
    w = o.bm1
    num = sum(np.sum(w * (x - y) ** 2) for x, y in zip(a[:2], b[:2]))
    den = sum(np.sum(w * y ** 2) for y in b[:2])
    return np.sqrt(num / den)


@pytest.mark.parametrize("mode,nsteps", [(0, 5), (1, 5)])
def test_steps_vs_oracle(hip6, oracle6, modes, mode, nsteps):
    """A few linearised steps with tight solver tolerances: GPU == oracle to ~1e-10
    (velocity, mass-weighted L2) and pressure to 1e-7."""
    o = oracle6
    q = _mode(o, modes, "dRe")
    hip6.set_tolerances(1e-13, 1e-13, 0)
    hip6.set_nsteps(nsteps)
    vq, vf = hip6.alloc(2)
    hip6.upload(vq, *q)
    hip6.matvec(vf, vq, mode)
    f = hip6.download(vf)
    ref = o.matvec(q, adjoint=bool(mode), nsteps=nsteps)
    hip6.set_nsteps(100)
    print("stats", hip6.stats())
    assert relL2(o, f, ref) < 1e-9
    assert np.abs(f[2] - ref[2]).max() / np.abs(ref[2]).max() < 1e-5
    hip6.free([vq, vf])
