// Symmetric tridiagonal eigen-solvers on the host (ai_tridiag.hip).  T has diagonal a[0..m) and off-diagonals b[1..m).
#pragma once
#include <vector>

// number of eigenvalues of T below x
int ai_sturm_lt_host(const double* a, const double* b, int m, double x);
// largest eigenvalue (theta_out) and its unit eigenvector s; `hint` (optional) = a value within 1e-13 relative of it
void ai_tridiag_top(const double* a, const double* b, int m, const double* hint, double* theta_out, std::vector<double>& s);
// idx-th largest eigenvalue (idx = 0: the largest) inside [lo, hi]
double ai_tridiag_eigval(const double* a, const double* b, int m, int idx, double lo, double hi);
// unit eigenvector x of the (already located) eigenvalue theta, kept orthogonal to prev[c] for c in cluster
void ai_tridiag_eigvec(const double* a, const double* b, int m, double theta, const std::vector<std::vector<double>>& prev,
                       const std::vector<int>& cluster, std::vector<double>& x);

