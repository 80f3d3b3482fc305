"""Host code of the recursion's harvest waves (`csrc/ai_tridiag.hip`, compiled here with g++; no GPU): the top eigenpair of a Lanczos
T_m, and the second coefficient vector of round 5's warm start -- inverse iteration with a shift just below theta, kept orthogonal to
the top eigenvector (`Flow::ritz_job` in `csrc/ai_flow.inc` calls exactly this)."""
import os
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_top_pair_and_the_warm_start_vector_of_a_tridiagonal_matrix(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(textwrap.dedent(r'''
        #include "ai_tridiag.h"
        #include <math.h>
        #include <cstdio>
        #include <random>
        static double rq(const std::vector<double>& a, const std::vector<double>& b, const std::vector<double>& x) {
          const int m = (int)a.size();
          double t = 0;
          for (int i = 0; i < m; ++i) {
            double y = a[i] * x[i];
            if (i > 0) y += b[i] * x[i - 1];
            if (i + 1 < m) y += b[i + 1] * x[i + 1];
            t += x[i] * y;
          }
          return t;
        }
        int main() {
          std::mt19937_64 rng(7);
          std::uniform_real_distribution<double> U(0.0, 1.0);
          double worst_res = 0, worst_orth = 0;
          for (int m : {2, 3, 8, 40, 150, 600}) for (int rep = 0; rep < 3; ++rep) {
            // a Lanczos-like T: diagonal around 0.9, off-diagonals 0.05 .. 0.3 (b[0] unused)
            std::vector<double> a(m), b(m, 0.0);
            for (int i = 0; i < m; ++i) a[i] = 0.85 + 0.1 * U(rng);
            for (int i = 1; i < m; ++i) b[i] = 0.05 + 0.25 * U(rng);
            double theta = 0;
            std::vector<double> s;
            ai_tridiag_top(a.data(), b.data(), m, nullptr, &theta, s);
            // (i) the top pair: residual of T s = theta s, and theta is the largest eigenvalue
            double res = 0;
            for (int i = 0; i < m; ++i) {
              double y = (a[i] - theta) * s[i];
              if (i > 0) y += b[i] * s[i - 1];
              if (i + 1 < m) y += b[i + 1] * s[i + 1];
              res = fmax(res, fabs(y));
            }
            worst_res = fmax(worst_res, res);
            if (ai_sturm_lt_host(a.data(), b.data(), m, theta + 1e-12) != m) return 2;
            if (ai_sturm_lt_host(a.data(), b.data(), m, theta - 1e-12) != m - 1) return 3;
            // (ii) the warm-start vector, as ritz_job forms it
            std::vector<std::vector<double>> prev(1, s);
            const std::vector<int> cluster{0};
            std::vector<double> x, x2;
            ai_tridiag_eigvec(a.data(), b.data(), m, theta - 1e-4 * fmax(1.0, fabs(theta)), prev, cluster, x);
            ai_tridiag_eigvec(a.data(), b.data(), m, theta - 1e-4 * fmax(1.0, fabs(theta)), prev, cluster, x2);
            double nn = 0, dot = 0;
            for (int i = 0; i < m; ++i) {
              nn += x[i] * x[i];
              dot += x[i] * s[i];
              if (x[i] != x2[i]) return 4;   // a function of T alone: the same bits again
            }
            if (fabs(nn - 1.0) > 1e-12) return 5;
            worst_orth = fmax(worst_orth, fabs(dot));
            // its Rayleigh quotient lies at the top of the REST of the spectrum: between the 5th- and the 2nd-largest eigenvalue
            const double lo = -10.0;
            const double t2 = ai_tridiag_eigval(a.data(), b.data(), m, 1, lo, theta);
            const double t5 = (m >= 5) ? ai_tridiag_eigval(a.data(), b.data(), m, 4, lo, theta) : lo;
            const double q = rq(a, b, x);
            if (!(q <= t2 + 1e-9 && q >= t5 - 1e-9)) { printf("m %d rq %.12f t2 %.12f t5 %.12f\n", m, q, t2, t5); return 6; }
          }
          printf("%.3e %.3e\n", worst_res, worst_orth);
          return (worst_res < 1e-12 && worst_orth < 1e-12) ? 0 : 7;
        }
    '''))
    exe = tmp_path / "t"
    csrc = os.path.join(ROOT, "autoinst_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", csrc, "-x", "c++", os.path.join(csrc, "ai_tridiag.hip"), str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
