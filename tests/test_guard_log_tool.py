"""tools/guard_log_diff.py reads the AI_FLOW_GUARD_LOG format of csrc/ai_flow.inc (guard_dump): a fabricated log of one rejected
Ritz pair and its repeat must come out with the first differing history row named.  (CPU only; the format is produced on the GPU by
tests/test_gpu_parity.py::test_a_spoiled_ritz_pair_is_caught_by_the_true_residual_and_solved_again when the variable is set.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rec(tag, m, restarts, slot, J, true, a):
    h = lambda v: float(v).hex()
    head = (f"{tag} g0 10 n 100 m {m} restarts {restarts} slot {slot} chunk 0 par 1 J {J} wave_mp {m} dev_theta {h(1.0)} dev_resid {h(2.0 ** -40)} "
            f"host_theta {h(1.0)} cu {h(0.0)} est {h(2.0 ** -40)} true {h(true)}\n")
    rows = {"a": a, "b": [1.0, 0.5, 0.25, 0.125][:m], "g": [0.0] * m, "c": [1.0, 0.5, 0.25, 0.125][:m]}
    return head + "".join(k + " " + " ".join(h(x) for x in v) + "\n" for k, v in rows.items())


def test_guard_log_diff_names_the_first_row_that_differs(tmp_path):
    log = tmp_path / "guard.log"
    log.write_text(_rec("BAD", 3, 0, 2, 50, 0.125, [-2.0 ** -58, 0.5, 0.25]) + "HISTROW wave_rows 3 row 1 mp 3 size device_m 36 host_m 32 attempt 1 J 9 g0 1 n 2\nhead 0x1p+0\n"
                   + _rec("REPEAT", 3, 1, 3, 90, 2.0 ** -36, [0.40625, 0.5, 0.25]))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "guard_log_diff.py"), str(log)], capture_output=True, text=True, check=True).stdout
    assert "1 rejected pair(s), 1 repeat(s)" in out
    assert "alpha: first difference at row 0" in out and "b: identical over the common rows" in out
    assert "Ritz coefficients: identical" in out
