"""Would a Gram-tile (matrix-core) form of the feature block keep parity?  Measured answer: yes on this data.

An MFMA formulation needs ||a - b||^2 = ||a||^2 + ||b||^2 - 2 a.b (v_mfma_f64_16x16x4 gives the a.b tile), while the
reference subtracts first (scipy cdist, ncuts_utils.py:130,144).  The expansion cancels when a ~ b; this script measures
the relative error of the resulting affinity factor exp(-w * ||a - b||) against the subtract-first form on the edges of
a synthetic chunk, in float64 throughout (CPU only).  On the synthetic features (0.3 noise per dimension, distances
of a few units) the error is 1e-15, so parity is NOT the reason the block stays off the matrix cores; the reason is the
roofline: 2 E F = 7 GFLOP of float64 for the 200k-point tri-modal chunk is ~0.09 ms at the vector rate against 2.2 ms
measured -- the kernel is bound by bringing the neighbours' feature rows to the lanes, which a Gram tile needs as well
(and it would compute all ~115 candidates per point instead of the ~37 inside the radius).  DESIGN.md section 5.

    python tests/tools/mfma_gram_error.py [n]  -> one JSON line (profiles/r02_mfma_gram_error.json)
"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from autoinst_amd import synth
from oracle import ncuts_ref

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
ch = synth.synthetic_chunk(n, 0, tarl=True, dino=True)
A = ncuts_ref.affinity_sparse(ch["points"], alpha=1.0).tocoo()
i, j = A.row, A.col
keep = i != j
i, j = i[keep], j[keep]
out = {"n": n, "edges": int(i.size)}
for name, F, w in (("tarl96", ch["tarl"], 0.5), ("dino384", ch["dino"], 0.1)):
    d_ref = ncuts_ref._rowwise_euclid(F, i, j)
    sq = np.einsum("ij,ij->i", F, F)
    dot = np.einsum("ij,ij->i", F[i], F[j])
    d2 = sq[i] + sq[j] - 2.0 * dot
    d_gram = np.sqrt(np.maximum(d2, 0.0))
    f_ref, f_gram = np.exp(-w * d_ref), np.exp(-w * d_gram)
    rel = np.abs(f_gram - f_ref) / f_ref
    nz = d_ref > 0
    out[name] = {"max_rel_err_of_factor": float(rel.max()), "edges_above_1e-12": int((rel > 1e-12).sum()),
                 "max_rel_err_of_distance": float((np.abs(d_gram - d_ref)[nz] / d_ref[nz]).max()),
                 "negative_d2": int((d2 < 0).sum()), "identical_rows_with_nonzero_gram_distance": int(((d_ref == 0) & (d_gram != 0)).sum())}
print(json.dumps(out))
