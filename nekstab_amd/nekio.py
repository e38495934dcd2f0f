"""Readers/writers for the Nek5000 on-disk formats that sit either side of the
hot path: ``.re2`` mesh, ``.ma2`` vertex map, ``.f0000N``
field files, and nekStab's ``Spectre_*.dat`` tables
(reference writer: core/eigensolvers.f:572-604; reader of fields: core/IO.f:15-60).

Host-side I/O only; nothing here is on the device path.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

_ENDIAN_TAG = 6.54321


@dataclass
class Re2Mesh:
    ndim: int
    nel: int
    xc: np.ndarray          # (nel, 2**ndim) vertex x, Nek "preprocessor" corner order
    yc: np.ndarray
    zc: np.ndarray | None
    curves: list            # (iel0, iface0, params[5], ctype)
    bcs: list               # (iel0, iface0, params[5], code) for field 1 (velocity)


def _endian(tag_bytes: bytes) -> str:
    if abs(struct.unpack("<f", tag_bytes)[0] - _ENDIAN_TAG) < 1e-3:
        return "<"
    if abs(struct.unpack(">f", tag_bytes)[0] - _ENDIAN_TAG) < 1e-3:
        return ">"
    raise ValueError("bad endian tag")


def read_re2(path: str) -> Re2Mesh:
    with open(path, "rb") as f:
        hdr = f.read(80).decode("ascii")
        if not (hdr.startswith("#v002") or hdr.startswith("#v003")):     # both: 8-byte reals; v003 = same record layout
            raise ValueError(f"unsupported re2 version: {hdr[:5]!r}")
        toks = hdr.split()
        nel, ndim = int(toks[1]), int(toks[2])
        en = _endian(f.read(4))
        nv = 2 ** ndim
        rec = 1 + ndim * nv
        raw = np.frombuffer(f.read(8 * rec * nel), dtype=en + "f8").reshape(nel, rec)
        xc = raw[:, 1:1 + nv].copy()
        yc = raw[:, 1 + nv:1 + 2 * nv].copy()
        zc = raw[:, 1 + 2 * nv:1 + 3 * nv].copy() if ndim == 3 else None
        ncurve = int(np.frombuffer(f.read(8), dtype=en + "f8")[0])
        curves = []
        for _ in range(ncurve):
            b = f.read(64)
            v = np.frombuffer(b[:56], dtype=en + "f8")
            curves.append((int(v[0]) - 1, int(v[1]) - 1, v[2:7].copy(),
                           b[56:64].decode("ascii").strip()))
        nbc = int(np.frombuffer(f.read(8), dtype=en + "f8")[0])
        bcs = []
        for _ in range(nbc):
            b = f.read(64)
            v = np.frombuffer(b[:56], dtype=en + "f8")
            bcs.append((int(v[0]) - 1, int(v[1]) - 1, v[2:7].copy(),
                        b[56:64].decode("ascii").strip()))
    return Re2Mesh(ndim, nel, xc, yc, zc, curves, bcs)


def read_ma2(path: str):
    """Returns (vertex ids (nel, 2**ndim), 1-based, lexicographic (r,s) corner
    order; partition-leaf id per element)."""
    with open(path, "rb") as f:
        hdr = f.read(132).decode("ascii")
        if not hdr.startswith("#v001"):
            raise ValueError("unsupported ma2 version")
        toks = hdr.split()
        nel, npts = int(toks[1]), int(toks[5])
        nv = npts // nel
        en = _endian(f.read(4))
        raw = np.frombuffer(f.read(4 * (nv + 1) * nel), dtype=en + "i4").reshape(nel, nv + 1)
    return raw[:, 1:].astype(np.int64), raw[:, 0].astype(np.int64)


@dataclass
class NekField:
    wdsize: int
    nx: int
    ny: int
    nz: int
    nel: int
    time: float
    istep: int
    rdcode: str
    elmap: np.ndarray                      # global element ids, 1-based, file order
    x: np.ndarray | None = None            # (ndim, nel, nz, ny, nx) in GLOBAL element order
    u: np.ndarray | None = None
    p: np.ndarray | None = None
    t: np.ndarray | None = None
    extra: dict = field(default_factory=dict)


def read_fld(path: str) -> NekField:
    """Field file -> arrays re-ordered to global element order (element ids in the
    file header list are 1-based and unsorted)."""
    with open(path, "rb") as f:
        hdr = f.read(132).decode("ascii")
        toks = hdr.split()
        if toks[0] != "#std":
            raise ValueError("not a Nek field file")
        wd, nx, ny, nz, nel = (int(t) for t in toks[1:6])
        time, istep = float(toks[7]), int(toks[8])
        rdcode = toks[11]
        en = _endian(f.read(4))
        elmap = np.frombuffer(f.read(4 * nel), dtype=en + "i4").astype(np.int64)
        ndim = 3 if nz > 1 else 2
        npt = nx * ny * nz
        dt = en + ("f8" if wd == 8 else "f4")
        order = np.argsort(elmap)
        out = NekField(wd, nx, ny, nz, nel, time, istep, rdcode, elmap)

        def vec(ncomp):
            a = np.frombuffer(f.read(wd * ncomp * npt * nel), dtype=dt)
            a = a.reshape(nel, ncomp, nz, ny, nx)[order].astype(np.float64)
            return np.ascontiguousarray(a.transpose(1, 0, 2, 3, 4))

        i = 0
        while i < len(rdcode):
            c = rdcode[i]
            if c == "X":
                out.x = vec(ndim)
            elif c == "U":
                out.u = vec(ndim)
            elif c == "P":
                out.p = vec(1)[0]
            elif c == "T":
                out.t = vec(1)[0]
            elif c == "S":
                ns = int(rdcode[i + 1:i + 3])
                out.extra["S"] = [vec(1)[0] for _ in range(ns)]
                i += 2
            i += 1
    return out


def write_fld(path: str, *, x=None, u=None, p=None, t=None, time=0.0, istep=0,
              wdsize=8):
    """Write a single-file Nek field (the `outpost` format nekStab's
    post-processing reads). Arrays are (ncomp, nel, nz, ny, nx) / (nel, nz, ny, nx)."""
    ref = x if x is not None else u
    if ref is None:
        ref = (p if p is not None else t)[None]
    _, nel, nz, ny, nx = ref.shape
    code = ("X" if x is not None else "") + ("U" if u is not None else "") + \
           ("P" if p is not None else "") + ("T" if t is not None else "")
    hdr = "#std %1d %2d %2d %2d %10d %10d %20.13E %9d %6d %6d %-10s %14.7E F" % (
        wdsize, nx, ny, nz, nel, nel, time, istep, 0, 1, code, 1.0)
    dt = "<f8" if wdsize == 8 else "<f4"
    with open(path, "wb") as f:
        f.write(hdr.ljust(132).encode("ascii"))
        f.write(struct.pack("<f", _ENDIAN_TAG))
        f.write(np.arange(1, nel + 1, dtype="<i4").tobytes())
        for a, isvec in ((x, True), (u, True), (p, False), (t, False)):
            if a is None:
                continue
            a = a if isvec else a[None]
            f.write(np.ascontiguousarray(a.transpose(1, 0, 2, 3, 4)).astype(dt).tobytes())


def read_spectre(path: str) -> np.ndarray:
    """Spectre_H*/NS* tables: rows of (Re, Im[, residual]) in E15.7."""
    return np.loadtxt(path, ndmin=2)


def write_spectre(path: str, vals: np.ndarray, residual: np.ndarray | None = None):
    """(3E15.7) / (2E15.7) rows, as core/eigensolvers.f:590-604 writes them."""
    def e157(v):
        # Fortran E15.7: 0.dddddddE+ee
        if v == 0.0 or not np.isfinite(v):
            return "%15s" % ("0.0000000E+00" if v == 0.0 else str(v))
        ex = int(np.floor(np.log10(abs(v)))) + 1
        m = v / 10.0 ** ex
        if abs(round(m, 7)) >= 1.0:
            m /= 10.0
            ex += 1
        s = "%.7f" % m
        s = s.replace("0.", "0.", 1)
        return "%15s" % ("%sE%+03d" % (s, ex))
    with open(path, "w") as f:
        for i, v in enumerate(vals):
            row = e157(v.real) + e157(v.imag)
            if residual is not None:
                row += e157(float(residual[i]))
            f.write(row + "\n")
