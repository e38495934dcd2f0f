"""1-D Legendre quadrature / Lagrange bases used by the host-side case
preparation (geometry generation, base-flow interpolation between orders).
The device library builds its own copies in C++ (csrc/basis.hpp)."""
from __future__ import annotations

import numpy as np


def _legendre(n: int, x: np.ndarray):
    """P_n(x) and P_n'(x) by the three-term recurrence."""
    p0 = np.ones_like(x)
    if n == 0:
        return p0, np.zeros_like(x)
    p1 = x.copy()
    for k in range(2, n + 1):
        p0, p1 = p1, ((2 * k - 1) * x * p1 - (k - 1) * p0) / k
    dp = n * (x * p1 - p0) / (x * x - 1.0 + (np.abs(x) == 1.0))
    return p1, dp


def gauss_legendre(n: int):
    """n Gauss-Legendre nodes (ascending) and weights on [-1,1]."""
    x = -np.cos(np.pi * (np.arange(n) + 0.75) / (n + 0.5))
    for _ in range(100):
        p, dp = _legendre(n, x)
        dx = p / dp
        x = x - dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    p, dp = _legendre(n, x)
    w = 2.0 / ((1.0 - x * x) * dp * dp)
    return x, w


def gauss_lobatto_legendre(n: int):
    """n Gauss-Lobatto-Legendre nodes (ascending) and weights on [-1,1]."""
    N = n - 1
    x = -np.cos(np.pi * np.arange(n) / N)
    xi = x[1:-1].copy()
    for _ in range(100):
        # roots of P_N'(x): Newton on q = P_N', q' from Legendre ODE
        p, dp = _legendre(N, xi)
        d2p = (2 * xi * dp - N * (N + 1) * p) / (1.0 - xi * xi)
        dx = dp / d2p
        xi = xi - dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    x[1:-1] = xi
    p, _ = _legendre(N, x)
    w = 2.0 / (N * (N + 1) * p * p)
    return x, w


def bary_weights(x: np.ndarray) -> np.ndarray:
    d = x[:, None] - x[None, :]
    np.fill_diagonal(d, 1.0)
    return 1.0 / np.prod(d, axis=1)


def interp_matrix(x_from: np.ndarray, x_to: np.ndarray) -> np.ndarray:
    """J[i,j] = l_j(x_to[i]) for the Lagrange basis on x_from."""
    w = bary_weights(x_from)
    J = np.zeros((len(x_to), len(x_from)))
    for i, xt in enumerate(x_to):
        d = xt - x_from
        hit = np.where(np.abs(d) < 1e-15)[0]
        if len(hit):
            J[i, hit[0]] = 1.0
        else:
            t = w / d
            J[i] = t / t.sum()
    return J


def deriv_matrix(x: np.ndarray) -> np.ndarray:
    """D[i,j] = l_j'(x[i]) on the nodes x themselves."""
    w = bary_weights(x)
    n = len(x)
    D = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if i != j:
                D[i, j] = (w[j] / w[i]) / (x[i] - x[j])
        D[i, i] = -np.sum(D[i])
    return D
