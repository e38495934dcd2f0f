// Host-side 1-D spectral bases for the device library: Gauss-Lobatto-Legendre
// (velocity mesh), Gauss-Legendre (pressure and dealiasing meshes), Lagrange
// derivative and interpolation matrices.  [UPSTREAM Nek5000 speclib: zwgll, zwgl,
// dgll, igllm] -- restated from the published formulas, not from Nek source.
#pragma once
#include <cmath>
#include <vector>

namespace nsk {

inline void legendre(int n, double x, double& p, double& dp) {
  double p0 = 1.0, p1 = x;
  if (n == 0) { p = 1.0; dp = 0.0; return; }
  for (int k = 2; k <= n; ++k) {
    double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
    p0 = p1; p1 = p2;
  }
  p = p1;
  dp = n * (x * p1 - p0) / (x * x - 1.0);
}

// n Gauss-Legendre nodes (ascending) and weights
inline void zwgl(int n, std::vector<double>& z, std::vector<double>& w) {
  z.resize(n); w.resize(n);
  for (int i = 0; i < n; ++i) {
    double x = -std::cos(M_PI * (i + 0.75) / (n + 0.5));
    double p, dp;
    for (int it = 0; it < 100; ++it) {
      legendre(n, x, p, dp);
      double dx = p / dp;
      x -= dx;
      if (std::fabs(dx) < 1e-16) break;
    }
    legendre(n, x, p, dp);
    z[i] = x;
    w[i] = 2.0 / ((1.0 - x * x) * dp * dp);
  }
  for (int i = 0; i < n / 2; ++i) {       // enforce exact symmetry
    double a = 0.5 * (z[n - 1 - i] - z[i]);
    z[i] = -a; z[n - 1 - i] = a;
    double b = 0.5 * (w[i] + w[n - 1 - i]);
    w[i] = w[n - 1 - i] = b;
  }
  if (n % 2) z[n / 2] = 0.0;
}

// n Gauss-Lobatto-Legendre nodes (ascending) and weights
inline void zwgll(int n, std::vector<double>& z, std::vector<double>& w) {
  const int N = n - 1;
  z.resize(n); w.resize(n);
  z[0] = -1.0; z[N] = 1.0;
  for (int i = 1; i < N; ++i) {
    double x = -std::cos(M_PI * i / N);
    for (int it = 0; it < 100; ++it) {
      double p, dp;
      legendre(N, x, p, dp);
      double d2p = (2 * x * dp - N * (N + 1) * p) / (1.0 - x * x);
      double dx = dp / d2p;
      x -= dx;
      if (std::fabs(dx) < 1e-16) break;
    }
    z[i] = x;
  }
  for (int i = 0; i < n / 2; ++i) {
    double a = 0.5 * (z[n - 1 - i] - z[i]);
    z[i] = -a; z[n - 1 - i] = a;
  }
  if (n % 2) z[n / 2] = 0.0;
  for (int i = 0; i < n; ++i) {
    double p, dp;
    if (i == 0 || i == N) { p = (i == 0 && (N % 2)) ? -1.0 : 1.0; }
    else legendre(N, z[i], p, dp);
    w[i] = 2.0 / (N * (N + 1) * p * p);
  }
}

// J[i*nf + j] = l_j(xt[i]) for the Lagrange basis on xf
inline std::vector<double> interp_matrix(const std::vector<double>& xf, const std::vector<double>& xt) {
  const int nf = (int)xf.size(), nt = (int)xt.size();
  std::vector<double> J((size_t)nt * nf);
  for (int i = 0; i < nt; ++i)
    for (int j = 0; j < nf; ++j) {
      double num = 1.0, den = 1.0;
      for (int k = 0; k < nf; ++k)
        if (k != j) { num *= (xt[i] - xf[k]); den *= (xf[j] - xf[k]); }
      J[(size_t)i * nf + j] = num / den;
    }
  return J;
}

// D[i*n + j] = l_j'(x[i])
inline std::vector<double> deriv_matrix(const std::vector<double>& x) {
  const int n = (int)x.size();
  std::vector<double> D((size_t)n * n, 0.0), c(n);
  for (int i = 0; i < n; ++i) {
    double p = 1.0;
    for (int k = 0; k < n; ++k) if (k != i) p *= (x[i] - x[k]);
    c[i] = p;
  }
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int j = 0; j < n; ++j)
      if (i != j) { D[(size_t)i * n + j] = c[i] / (c[j] * (x[i] - x[j])); s += D[(size_t)i * n + j]; }
    D[(size_t)i * n + i] = -s;
  }
  return D;
}

inline std::vector<double> matmul(const std::vector<double>& A, const std::vector<double>& B, int m, int k, int n) {
  std::vector<double> C((size_t)m * n, 0.0);
  for (int i = 0; i < m; ++i)
    for (int l = 0; l < k; ++l) {
      double a = A[(size_t)i * k + l];
      for (int j = 0; j < n; ++j) C[(size_t)i * n + j] += a * B[(size_t)l * n + j];
    }
  return C;
}

}  // namespace nsk
