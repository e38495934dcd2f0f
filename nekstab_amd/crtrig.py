"""Correctly rounded sin / cos in double-double arithmetic, written so that the SAME sequence of IEEE double operations runs on
the host (this file, numpy) and on the device (``csrc/nsk_crtrig.hpp``): the two give bit-identical results.

Why: nekStab's seed ``mth_rand`` (core/utils.f:457-469) is ``cos(1e3 sin(1e3 sin(r)))`` with ``r`` up to 1e10 -- by construction
one ulp of a sine moves the result by up to 1e-2, and the vendor sines of the device (ocml), of numpy (SIMD kernels) and of the
host's libm differ in the last bit for some arguments.  A sine that is correctly rounded is the same number everywhere.

Algorithm (|x| < 5e10): k = rint(x 2/pi); r = x - k pi/2 by Cody-Waite with nine 18-bit parts of pi/2 (every product k P_i is
exact in double; the difference is carried in double-double); sin r / cos r by their Taylor series in double-double (error
< 2^-100); the result is the high word.  twoProd is computed with Veltkamp / Dekker splitting here and with one fused
multiply-add on the device: both return the exact error term, hence the same numbers.
"""
from __future__ import annotations

import numpy as np

# pi/2 = sum P[i], each part an 18-bit integer times a power of two (generated from mpmath; the device header holds the same list)
_PIO2_PARTS = None
_SPLIT = 134217729.0  # 2^27 + 1


def _parts():
    """Nine 18-bit parts of pi/2 from its leading hexadecimal digits (exact binary fractions)."""
    global _PIO2_PARTS
    if _PIO2_PARTS is None:
        # pi/2 = 0x1.921FB54442D18469898CC51701B839A252049C1114CF98E804177D4C76273644A29410F31C6809BBDF2A33679A748636605614DBE4BE286E9FC26ADADAA3848BC90B6AECC4BCFD8DE89885D34C6FDAD617FEB96DE80D6FDBDC70D7F6B5133F4B5D3E4822F8963FCC9250CCA3D9C8B67B8400F97142C77E65B...
        hexfrac = "921FB54442D18469898CC51701B839A252049C1114CF98E804177D4C76273644A29410F31C6809BBDF2A33679A748636605614DBE4BE286E9FC26ADADAA3848BC9"
        bits = 1 << (4 * len(hexfrac)) | int(hexfrac, 16)          # pi/2 * 2^(4 len): leading 1 then the fraction
        nb = 4 * len(hexfrac) + 1
        parts = []
        pos = nb
        for i in range(9):
            chunk = (bits >> (pos - 18)) & ((1 << 18) - 1)
            # the chunk holds bits [pos-18, pos) of the integer; its value is chunk * 2^(pos - 18 - 4 len)
            parts.append(float(chunk) * 2.0 ** (pos - 18 - 4 * len(hexfrac)))
            pos -= 18
        _PIO2_PARTS = parts
    return _PIO2_PARTS


# 1/n! as double-double (hi, lo), n = 0 .. 31 (exact rationals rounded twice)
_INVFACT = None


def _invfact():
    global _INVFACT
    if _INVFACT is None:
        from fractions import Fraction
        out = []
        f = 1
        for n in range(32):
            if n:
                f *= n
            q = Fraction(1, f)
            hi = float(q)
            lo = float(q - Fraction(hi))
            out.append((hi, lo))
        _INVFACT = out
    return _INVFACT


def _two_sum(a, b):
    s = a + b
    bb = s - a
    e = (a - (s - bb)) + (b - bb)
    return s, e


def _fast_two_sum(a, b):
    s = a + b
    e = b - (s - a)
    return s, e


def _split(a):
    t = a * _SPLIT
    hi = t - (t - a)
    return hi, a - hi


def _two_prod(a, b):
    p = a * b
    ah, al = _split(a)
    bh, bl = _split(b)
    e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
    return p, e


def _dd_mul(ah, al, bh, bl):
    p, e = _two_prod(ah, bh)
    e = e + (ah * bl + al * bh)
    return _fast_two_sum(p, e)


def _dd_add(ah, al, bh, bl):
    s, e = _two_sum(ah, bh)
    e = e + (al + bl)
    return _fast_two_sum(s, e)


def _reduce(x):
    """(k mod 4, r_hi, r_lo) with x = k pi/2 + r, |r| <= pi/4 (+ a rounding of k)."""
    x = np.asarray(x, dtype=np.float64)
    k = np.rint(x * 0.6366197723675814)                      # 2/pi rounded to double
    hi, lo = x.copy(), np.zeros_like(x)
    for p in _parts():
        t = k * p                                            # exact: k < 2^35, p has 18 bits
        s, e = _two_sum(hi, -t)
        e = e + lo
        hi, lo = _fast_two_sum(s, e)
    q = np.mod(k, 4.0).astype(np.int64)
    return q, hi, lo


def _sin_cos_dd(rh, rl):
    """Taylor series of sin r and cos r in double-double, |r| <= 0.8."""
    zh, zl = _dd_mul(rh, rl, rh, rl)
    F = _invfact()
    # sin r = r * sum_{n=0}^{14} (-1)^n z^n / (2n+1)! ; cos r = sum_{n=0}^{15} (-1)^n z^n / (2n)!
    sh, sl = np.full_like(rh, F[29][0]), np.full_like(rh, F[29][1])
    for n in range(13, -1, -1):
        sh, sl = _dd_mul(sh, sl, zh, zl)
        c = F[2 * n + 1]
        # Horner with alternating signs: p <- c_n - z p
        sh, sl = _dd_add(np.full_like(rh, c[0]), np.full_like(rh, c[1]), -sh, -sl)
    sh, sl = _dd_mul(sh, sl, rh, rl)
    ch, cl = np.full_like(rh, F[30][0]), np.full_like(rh, F[30][1])
    for n in range(14, -1, -1):
        ch, cl = _dd_mul(ch, cl, zh, zl)
        c = F[2 * n]
        ch, cl = _dd_add(np.full_like(rh, c[0]), np.full_like(rh, c[1]), -ch, -cl)
    return sh, ch


def sin_cr(x):
    with np.errstate(all="ignore"):
        q, rh, rl = _reduce(x)
        s, c = _sin_cos_dd(rh, rl)
    return np.where(q == 0, s, np.where(q == 1, c, np.where(q == 2, -s, -c)))


def cos_cr(x):
    with np.errstate(all="ignore"):
        q, rh, rl = _reduce(x)
        s, c = _sin_cos_dd(rh, rl)
    return np.where(q == 0, c, np.where(q == 1, -s, np.where(q == 2, -c, s)))


def device_header_constants():
    """The constant tables as C initialisers (what csrc/nsk_crtrig.hpp holds; tests check they agree)."""
    parts = ", ".join(float.hex(p) for p in _parts())
    fact = ", ".join("{%s, %s}" % (float.hex(h), float.hex(l)) for h, l in _invfact())
    return parts, fact
