"""Compare every rejected Ritz pair in an AI_FLOW_GUARD_LOG file with the repeat of its segment, bit for bit:
    python tools/guard_log_diff.py gpurun_out/soakm/guard_a.log ...
Says which stage differed: the size of T at the freeze (the check), T itself (the Lanczos steps: SpMV / update), or only the
Ritz coefficients / the vector (host eigen-solve, upload, fk_ritz)."""
import sys


def parse(path):
    recs, cur = [], None
    for line in open(path):
        w = line.split()
        if not w:
            continue
        if w[0] in ("BAD", "REPEAT"):
            cur = {"tag": w[0]}
            for k, v in zip(w[1::2], w[2::2]):
                cur[k] = float.fromhex(v) if v.startswith(("0x", "-0x")) or "p" in v else int(v)
            recs.append(cur)
        elif w[0] in ("a", "b", "g", "c") and cur is not None:
            cur[w[0] + "_"] = [float.fromhex(x) for x in w[1:]]
    return recs


def first_diff(x, y):
    for i, (p, q) in enumerate(zip(x, y)):
        if p != q:
            return i
    return None


for path in sys.argv[1:]:
    recs = parse(path)
    bad = [r for r in recs if r["tag"] == "BAD"]
    print(f"{path}: {len(bad)} rejected pair(s), {len(recs) - len(bad)} repeat(s)")
    for i, r in enumerate(recs):
        if r["tag"] != "BAD":
            continue
        rep = next((q for q in recs[i + 1:] if q["tag"] == "REPEAT" and q["g0"] == r["g0"] and q["n"] == r["n"]), None)
        print(f"  segment {r['g0']}+{r['n']}: rejected at m = {r['m']} (J = {r['J']}, slot {r['slot']}, restarts {r['restarts']}): estimate {r['est']:.3g}, true {r['true']:.3g}, "
              f"theta device {r['dev_theta']!r} host {r['host_theta']!r}")
        if rep is None:
            print("    no repeat of this segment in the file")
            continue
        print(f"    repeat: m = {rep['m']} (J = {rep['J']}, slot {rep['slot']}): estimate {rep['est']:.3g}, true {rep['true']:.3g}, theta device {rep['dev_theta']!r} host {rep['host_theta']!r}")
        mm = min(r["m"], rep["m"])
        for k, what in (("a_", "alpha"), ("b_", "b"), ("g_", "g")):
            d = first_diff(r[k][:mm], rep[k][:mm])
            print(f"    {what}: " + ("identical over the common rows" if d is None else f"first difference at row {d}: {r[k][d]!r} vs {rep[k][d]!r}"))
        if r["m"] == rep["m"]:
            d = first_diff(r["c_"], rep["c_"])
            print("    Ritz coefficients: " + ("identical" if d is None else f"first difference at {d}: {r['c_'][d]!r} vs {rep['c_'][d]!r}"))
        else:
            print(f"    size of T differs: {r['m']} vs {rep['m']}")
