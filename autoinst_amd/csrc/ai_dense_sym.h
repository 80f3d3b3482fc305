// Dense symmetric eigenproblems of the projected (B x B, B <= 128) matrices of the Chebyshev-filtered subspace iteration: host
// code, no HIP in here (tests/test_host_eigh.py compiles this header with g++ and checks one solver against the other).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

// cyclic Jacobi: eigenvalues (descending) and eigenvectors (columns of q, row-major) of the symmetric a
inline void cf_jacobi(std::vector<double> a, int B, std::vector<double>& evals, std::vector<double>& q) {
  q.assign((size_t)B * B, 0.0);
  for (int i = 0; i < B; ++i) q[(size_t)i * B + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < B; ++i) {
      diag += a[(size_t)i * B + i] * a[(size_t)i * B + i];
      for (int j = i + 1; j < B; ++j) off += a[(size_t)i * B + j] * a[(size_t)i * B + j];
    }
    if (off <= 1e-32 * std::max(diag, 1e-300)) break;
    for (int p = 0; p < B - 1; ++p)
      for (int r = p + 1; r < B; ++r) {
        const double apr = a[(size_t)p * B + r];
        if (fabs(apr) < 1e-300) continue;
        const double app = a[(size_t)p * B + p], arr = a[(size_t)r * B + r];
        const double tau = (arr - app) / (2.0 * apr);
        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < B; ++k) {  // columns p, r
          const double akp = a[(size_t)k * B + p], akr = a[(size_t)k * B + r];
          a[(size_t)k * B + p] = c * akp - s * akr;
          a[(size_t)k * B + r] = s * akp + c * akr;
        }
        for (int k = 0; k < B; ++k) {  // rows p, r
          const double apk = a[(size_t)p * B + k], ark = a[(size_t)r * B + k];
          a[(size_t)p * B + k] = c * apk - s * ark;
          a[(size_t)r * B + k] = s * apk + c * ark;
        }
        for (int k = 0; k < B; ++k) {
          const double qkp = q[(size_t)k * B + p], qkr = q[(size_t)k * B + r];
          q[(size_t)k * B + p] = c * qkp - s * qkr;
          q[(size_t)k * B + r] = s * qkp + c * qkr;
        }
      }
  }
  std::vector<int> order(B);
  for (int i = 0; i < B; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int x, int y) { return a[(size_t)x * B + x] > a[(size_t)y * B + y]; });
  evals.resize(B);
  std::vector<double> qs((size_t)B * B);
  for (int j = 0; j < B; ++j) {
    evals[j] = a[(size_t)order[j] * B + order[j]];
    for (int i = 0; i < B; ++i) qs[(size_t)i * B + j] = q[(size_t)i * B + order[j]];
  }
  q.swap(qs);
}

// The same by Householder tridiagonalisation + implicit QL (the EISPACK tred2 / tql2 pair, as restated in JAMA): 1 ms for a
// 128 x 128 matrix where the cyclic Jacobi above takes 16 ms -- seven of them were 8 % of a cfg5 solve.  Eigenvalues descending,
// eigenvectors in the columns of q (row-major).  Returns false if an eigenvalue does not converge in 60 QL sweeps (the caller
// then falls back to Jacobi).
inline bool cf_eigh(const std::vector<double>& a, int n, std::vector<double>& evals, std::vector<double>& q) {
  std::vector<double> V(a), d(n), e(n);
  auto at = [&](int i, int j) -> double& { return V[(size_t)i * n + j]; };
  for (int j = 0; j < n; ++j) d[j] = at(n - 1, j);
  for (int i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (int k = 0; k < i; ++k) scale += fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (int j = 0; j < i; ++j) {
        d[j] = at(i - 1, j);
        at(i, j) = 0.0;
        at(j, i) = 0.0;
      }
    } else {
      for (int k = 0; k < i; ++k) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1], g = sqrt(h);
      if (f > 0.0) g = -g;
      e[i] = scale * g;
      h -= f * g;
      d[i - 1] = f - g;
      for (int j = 0; j < i; ++j) e[j] = 0.0;
      for (int j = 0; j < i; ++j) {
        f = d[j];
        at(j, i) = f;
        g = e[j] + at(j, j) * f;
        for (int k = j + 1; k <= i - 1; ++k) {
          g += at(k, j) * d[k];
          e[k] += at(k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (int j = 0; j < i; ++j) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (int j = 0; j < i; ++j) {
        f = d[j];
        g = e[j];
        for (int k = j; k <= i - 1; ++k) at(k, j) -= (f * e[k] + g * d[k]);
        d[j] = at(i - 1, j);
        at(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  for (int i = 0; i < n - 1; ++i) {
    at(n - 1, i) = at(i, i);
    at(i, i) = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (int k = 0; k <= i; ++k) d[k] = at(k, i + 1) / h;
      for (int j = 0; j <= i; ++j) {
        double g = 0.0;
        for (int k = 0; k <= i; ++k) g += at(k, i + 1) * at(k, j);
        for (int k = 0; k <= i; ++k) at(k, j) -= g * d[k];
      }
    }
    for (int k = 0; k <= i; ++k) at(k, i + 1) = 0.0;
  }
  for (int j = 0; j < n; ++j) {
    d[j] = at(n - 1, j);
    at(n - 1, j) = 0.0;
  }
  at(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // implicit QL; the plane rotations act on two COLUMNS of V: done on the rows of its transpose (contiguous)
  std::vector<double> Z((size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k) Z[(size_t)i * n + k] = at(k, i);
  for (int i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = ldexp(1.0, -52);
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, fabs(d[l]) + fabs(e[l]));
    int m = l;
    while (m < n - 1 && fabs(e[m]) > eps * tst1) ++m;
    if (m > l) {
      int iter = 0;
      do {
        if (++iter > 60) return false;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = hypot(p, 1.0);
        if (p < 0.0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
        const double el1 = e[l + 1];
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          double* z1 = &Z[(size_t)(i + 1) * n];
          double* z0 = &Z[(size_t)i * n];
          for (int k = 0; k < n; ++k) {
            h = z1[k];
            z1[k] = s * z0[k] + c * h;
            z0[k] = c * z0[k] - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (fabs(e[l]) > eps * tst1);
    }
    d[l] += f;
    e[l] = 0.0;
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return d[x] > d[y]; });
  evals.resize(n);
  q.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    evals[j] = d[order[j]];
    for (int i = 0; i < n; ++i) q[(size_t)i * n + j] = Z[(size_t)order[j] * n + i];
  }
  return true;
}

