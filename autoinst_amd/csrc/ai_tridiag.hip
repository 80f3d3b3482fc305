// Host-side eigen-solvers of the small symmetric tridiagonal matrices T_m the Lanczos recurrences produce
// (LAPACK is not linked): Sturm counts, bisection, pivoted-LU inverse iteration (as dstein).
#include <math.h>

#include <algorithm>
#include <utility>

#include "ai_tridiag.h"

// ----------------------------------------------------------------------------- host: tridiagonal eigenvector
// Eigenvector of the largest eigenvalue of T (diag a[0..m), off-diag b[1..m)): bisection for
// the eigenvalue, then inverse iteration with a pivoted tridiagonal solve (as LAPACK dstein).
int ai_sturm_lt_host(const double* a, const double* b, int m, double x) {
  int cnt = 0;
  double q = a[0] - x;
  if (q < 0) ++cnt;
  for (int i = 1; i < m; ++i) {
    if (fabs(q) < 1e-300) q = (q < 0) ? -1e-300 : 1e-300;
    q = a[i] - x - b[i] * b[i] / q;
    if (q < 0) ++cnt;
  }
  return cnt;
}

void ai_tridiag_top(const double* a, const double* b, int m, const double* hint, double* theta_out, std::vector<double>& s) {
  s.assign(m, 0.0);
  if (m == 1) {
    *theta_out = a[0];
    s[0] = 1.0;
    return;
  }
  double lo = -1e300, hi = -1e300, nrm = 0.0;
  for (int i = 0; i < m; ++i) {
    const double bl = i > 0 ? fabs(b[i]) : 0.0, br = i + 1 < m ? fabs(b[i + 1]) : 0.0;
    lo = std::max(lo, a[i]);
    hi = std::max(hi, a[i] + bl + br);
    nrm = std::max(nrm, fabs(a[i]) + bl + br);
  }
  lo -= 1e-14 * std::max(fabs(lo), 1.0);
  hi += 1e-14 * std::max(fabs(hi), 1.0);
  if (hint) {
    // the device check already located the eigenvalue: verify a tight bracket around it
    const double w = 1e-13 * std::max(fabs(*hint), 1.0);
    const double l2 = *hint - w, h2 = *hint + w;
    if (l2 > lo && ai_sturm_lt_host(a, b, m, l2) < m) lo = l2;
    if (h2 < hi && ai_sturm_lt_host(a, b, m, h2) == m) hi = h2;
  }
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid <= lo || mid >= hi) break;
    if (ai_sturm_lt_host(a, b, m, mid) == m) hi = mid; else lo = mid;
  }
  const double theta = 0.5 * (lo + hi);
  *theta_out = theta;
  // LU of (T - theta I) with partial pivoting (rows i, i+1): dl, d, du, du2
  std::vector<double> d(m), du(m, 0.0), du2(m, 0.0), dl(m, 0.0);
  std::vector<int> piv(m, 0);
  const double tiny = 2.3e-16 * std::max(nrm, 1e-300);
  for (int i = 0; i < m; ++i) d[i] = a[i] - theta;
  for (int i = 0; i + 1 < m; ++i) {
    du[i] = b[i + 1];
    dl[i] = b[i + 1];
  }
  for (int i = 0; i + 1 < m; ++i) {
    if (fabs(d[i]) >= fabs(dl[i])) {
      if (fabs(d[i]) < tiny) d[i] = tiny;
      const double f = dl[i] / d[i];
      dl[i] = f;
      d[i + 1] -= f * du[i];
      du2[i] = 0.0;
      piv[i] = 0;
    } else {
      const double f = d[i] / dl[i];
      d[i] = dl[i];
      dl[i] = f;
      const double t = du[i];
      du[i] = d[i + 1];
      d[i + 1] = t - f * du[i];
      if (i + 2 < m) {
        du2[i] = du[i + 1];
        du[i + 1] = -f * du[i + 1];
      }
      piv[i] = 1;
    }
  }
  if (fabs(d[m - 1]) < tiny) d[m - 1] = tiny;
  std::vector<double> x(m);
  for (int i = 0; i < m; ++i) x[i] = 1.0 + 0.001 * ((i * 2654435761u) % 1000) / 1000.0;  // fixed, generic start
  for (int iter = 0; iter < 4; ++iter) {
    for (int i = 0; i + 1 < m; ++i) {  // forward: L y = P x
      if (piv[i]) std::swap(x[i], x[i + 1]);
      x[i + 1] -= dl[i] * x[i];
    }
    x[m - 1] /= d[m - 1];  // backward: U z = y
    if (m >= 2) x[m - 2] = (x[m - 2] - du[m - 2] * x[m - 1]) / d[m - 2];
    for (int i = m - 3; i >= 0; --i) x[i] = (x[i] - du[i] * x[i + 1] - du2[i] * x[i + 2]) / d[i];
    double n2 = 0.0;
    for (int i = 0; i < m; ++i) n2 += x[i] * x[i];
    const double rn = 1.0 / sqrt(n2);
    for (int i = 0; i < m; ++i) x[i] *= rn;
  }
  s = x;
}

// idx-th largest eigenvalue (idx = 0: the largest) of T by bisection on Sturm counts
double ai_tridiag_eigval(const double* a, const double* b, int m, int idx, double lo, double hi) {
  const int need = m - idx;  // smallest x with count_lt(x) >= need is just above the wanted eigenvalue
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid <= lo || mid >= hi) break;
    if (ai_sturm_lt_host(a, b, m, mid) >= need) hi = mid; else lo = mid;
  }
  return 0.5 * (lo + hi);
}

// eigenvector of T for the (already located) eigenvalue theta: pivoted LU + inverse iteration,
// kept orthogonal to `prev` (eigenvectors of neighbouring eigenvalues, as LAPACK dstein does)
void ai_tridiag_eigvec(const double* a, const double* b, int m, double theta, const std::vector<std::vector<double>>& prev,
                           const std::vector<int>& cluster, std::vector<double>& x) {
  x.assign(m, 0.0);
  if (m == 1) {
    x[0] = 1.0;
    return;
  }
  double nrm = 0.0;
  for (int i = 0; i < m; ++i) nrm = std::max(nrm, fabs(a[i]) + (i > 0 ? fabs(b[i]) : 0.0) + (i + 1 < m ? fabs(b[i + 1]) : 0.0));
  std::vector<double> d(m), du(m, 0.0), du2(m, 0.0), dl(m, 0.0);
  std::vector<int> piv(m, 0);
  const double tiny = 2.3e-16 * std::max(nrm, 1e-300);
  for (int i = 0; i < m; ++i) d[i] = a[i] - theta;
  for (int i = 0; i + 1 < m; ++i) {
    du[i] = b[i + 1];
    dl[i] = b[i + 1];
  }
  for (int i = 0; i + 1 < m; ++i) {
    if (fabs(d[i]) >= fabs(dl[i])) {
      if (fabs(d[i]) < tiny) d[i] = tiny;
      const double f = dl[i] / d[i];
      dl[i] = f;
      d[i + 1] -= f * du[i];
      du2[i] = 0.0;
      piv[i] = 0;
    } else {
      const double f = d[i] / dl[i];
      d[i] = dl[i];
      dl[i] = f;
      const double t = du[i];
      du[i] = d[i + 1];
      d[i + 1] = t - f * du[i];
      if (i + 2 < m) {
        du2[i] = du[i + 1];
        du[i + 1] = -f * du[i + 1];
      }
      piv[i] = 1;
    }
  }
  if (fabs(d[m - 1]) < tiny) d[m - 1] = tiny;
  for (int i = 0; i < m; ++i) x[i] = 1.0 + 0.001 * ((i * 2654435761u) % 1000) / 1000.0;
  for (int iter = 0; iter < 5; ++iter) {
    for (int c : cluster) {  // stay orthogonal to the eigenvectors of the cluster found so far
      double dot = 0.0;
      for (int i = 0; i < m; ++i) dot += prev[c][i] * x[i];
      for (int i = 0; i < m; ++i) x[i] -= dot * prev[c][i];
    }
    for (int i = 0; i + 1 < m; ++i) {
      if (piv[i]) std::swap(x[i], x[i + 1]);
      x[i + 1] -= dl[i] * x[i];
    }
    x[m - 1] /= d[m - 1];
    if (m >= 2) x[m - 2] = (x[m - 2] - du[m - 2] * x[m - 1]) / d[m - 2];
    for (int i = m - 3; i >= 0; --i) x[i] = (x[i] - du[i] * x[i + 1] - du2[i] * x[i + 2]) / d[i];
    double n2 = 0.0;
    for (int i = 0; i < m; ++i) n2 += x[i] * x[i];
    const double rn = 1.0 / sqrt(n2);
    for (int i = 0; i < m; ++i) x[i] *= rn;
  }
  for (int c : cluster) {
    double dot = 0.0;
    for (int i = 0; i < m; ++i) dot += prev[c][i] * x[i];
    for (int i = 0; i < m; ++i) x[i] -= dot * prev[c][i];
  }
  double n2 = 0.0;
  for (int i = 0; i < m; ++i) n2 += x[i] * x[i];
  const double rn = 1.0 / sqrt(n2);
  for (int i = 0; i < m; ++i) x[i] *= rn;
}
