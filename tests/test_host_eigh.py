"""Host code of the cfg5 solver: the QL eigensolver of the projected 128 x 128 problems against the Jacobi solver it replaced
(`csrc/ai_dense_sym.h`, compiled here with g++; no GPU)."""
import os, subprocess, textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ql_eigensolver_matches_jacobi(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(textwrap.dedent(r'''
        #include "ai_dense_sym.h"
        #include <cstdio>
        #include <random>
        int main() {
          std::mt19937_64 rng(1);
          std::normal_distribution<double> N(0, 1);
          double wd = 0, wo = 0, wr = 0;
          for (int n : {1, 2, 3, 5, 64, 128}) for (int rep = 0; rep < 4; ++rep) {
            std::vector<double> a((size_t)n * n);
            for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) {
              double v = N(rng);
              if (rep == 1 && i != j) v *= 1e-9;          // nearly diagonal
              if (rep == 2) v = (i == j) ? 1.0 : 0.0;      // identity: every eigenvalue equal
              if (rep == 3 && j > i + 1) v = 0;            // already tridiagonal
              a[(size_t)i * n + j] = a[(size_t)j * n + i] = v;
            }
            std::vector<double> e1, q1, e2, q2;
            cf_jacobi(a, n, e1, q1);
            if (!cf_eigh(a, n, e2, q2)) return 1;
            for (int i = 0; i < n; ++i) wd = std::max(wd, fabs(e1[i] - e2[i]));
            for (int i = 0; i + 1 < n; ++i) if (e2[i] < e2[i + 1]) return 3;   // descending
            for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
              double t = 0; for (int k = 0; k < n; ++k) t += q2[(size_t)k * n + i] * q2[(size_t)k * n + j];
              wo = std::max(wo, fabs(t - (i == j)));
            }
            for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
              double t = 0; for (int k = 0; k < n; ++k) t += a[(size_t)i * n + k] * q2[(size_t)k * n + j];
              wr = std::max(wr, fabs(t - e2[j] * q2[(size_t)i * n + j]));
            }
          }
          printf("%.3e %.3e %.3e\\n", wd, wo, wr);
          return (wd < 1e-11 && wo < 1e-12 && wr < 1e-12) ? 0 : 2;
        }
    '''))
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "autoinst_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
