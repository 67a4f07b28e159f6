"""Generates the committed golden fixtures under tests/golden/.

The reference (dibbelab/gficf) holds no tests / golden vectors for this path and cannot be
run or compiled in this image (no R; its Jaccard translation unit needs Rcpp headers), so
these vectors are NOT reference outputs.  They are:
  * ``known_answers.json`` — hand-derived known answers (worked by hand from the
    reference's formulas; see the comments in this file), and
  * ``jaccard_cases.npz`` / ``gficf_cases.npz`` — regression vectors produced by the CPU
    oracle (oracle/, two independent restatements that agree), inputs + expected outputs.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from gficf_amd import synth  # noqa: E402
from oracle import oracle_np  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def known_answers():
    # Jaccard toy, 5 cells x k=3 (1-based ids), worked by hand:
    #   row1 {2,3,4} row2 {1,3,5} row3 {1,2,4} row4 {3,5,1} row5 {4,1,2}
    #   edge (1->2): {2,3,4}∩{1,3,5} = {3}     u=1 -> 1/(6-1) = 0.2
    #   edge (1->3): {2,3,4}∩{1,2,4} = {2,4}   u=2 -> 2/(6-2) = 0.5
    #   edge (1->4): {2,3,4}∩{1,3,5} = {3}     u=1 -> 0.2   ... (see u table)
    toy = [[2, 3, 4], [1, 3, 5], [1, 2, 4], [3, 5, 1], [4, 1, 2]]
    toy_u = [[1, 2, 1], [1, 1, 1], [2, 1, 1], [1, 1, 1], [1, 2, 1]]
    # self id inside the own row (weight 1 when both rows are equal sets: u = k -> k/(2k-k) = 1)
    selfloop = [[1, 2], [1, 2], [1, 2]]          # every row {1,2}; all u = 2 -> weight 1.0
    # edges with empty intersection (u = 0 -> zero row), worked by hand:
    #   r1 {2,3} r2 {4,5} r3 {1,2} r4 {1,3} r5 {1,4}
    #   (1->2) {2,3}∩{4,5}=∅ 0   (1->3) {2,3}∩{1,2}={2} 1
    #   (2->4) {4,5}∩{1,3}=∅ 0   (2->5) {4,5}∩{1,4}={4} 1
    #   (3->1) {1,2}∩{2,3}={2} 1 (3->2) {1,2}∩{4,5}=∅ 0
    #   (4->1) {1,3}∩{2,3}={3} 1 (4->3) {1,3}∩{1,2}={1} 1
    #   (5->1) {1,4}∩{2,3}=∅ 0   (5->4) {1,4}∩{1,3}={1} 1
    disjoint = [[2, 3], [4, 5], [1, 2], [1, 3], [1, 4]]
    disjoint_u = [[0, 1], [0, 1], [1, 0], [1, 1], [0, 1]]
    # duplicates inside rows (multiset semantics of std::set_intersection):
    #   row1 (2,2,3) row2 (2,2,2) row3 (1,1,2)
    #   (1->2 twice): {2,2,3}∩{2,2,2} = {2,2} u=2;  (1->3): {2,2,3}∩{1,1,2} = {2} u=1
    #   (2->2 x3):    {2,2,2}∩{2,2,2} u=3
    #   (3->1 twice): {1,1,2}∩{2,2,3} = {2} u=1;   (3->2): {1,1,2}∩{2,2,2} = {2} u=1
    dups = [[2, 2, 3], [2, 2, 2], [1, 1, 2]]
    dups_u = [[2, 2, 1], [3, 3, 3], [1, 1, 1]]
    # GF-ICF 4 genes x 3 cells, min=0, max=1:
    #   S = [4, 8, 4]; nt = [2,2,3,1]; w = [ln(4/3), ln(4/3), 0, ln 2]
    M = [[1, 0, 3], [1, 2, 0], [2, 2, 1], [0, 4, 0]]
    gf = {
        "M": M, "min": 0.0, "max": 1.0,
        "nt": [2, 2, 3, 1],
        "w": [0.28768207245178085, 0.28768207245178085, 0.0, 0.6931471805599453],
        "dense": [[0.7071067811865475, 0.0, 1.0],
                  [0.7071067811865475, 0.20318977863036333, 0.0],
                  [0.0, 0.0, 0.0],
                  [0.0, 0.9791393740730396, 0.0]],
    }
    # filter case: same M, min = 0.5 -> keep nt > 1.5: genes 0,1,2 (nt 2,2,3); gene 3 dropped.
    #   S = [4, 4, 4]; w over kept = [ln(4/3), ln(4/3), 0]
    #   c1: tf = [.25,.25,.5]*w = [a,a,0] -> l2 -> [1/sqrt2, 1/sqrt2, 0]
    #   c2: tf = [0,.5,.5] -> [0, .5*ln(4/3), 0] -> [0,1,0];  c3: [.75,0,.25] -> [1,0,0]
    gf_filter = {
        "M": M, "min": 0.5, "max": 1.0, "keep": [1, 1, 1, 0],
        "dense": [[0.7071067811865475, 0.0, 1.0], [0.7071067811865475, 1.0, 0.0], [0.0, 0.0, 0.0]],
    }
    # a cell whose only kept gene has w = 0 -> q = 0 -> 1/sqrt(0) = Inf -> 0 (R/gficf.R:101): all zeros
    gf_w0 = {"M": [[1, 2], [0, 3]], "min": 0.0, "max": 1.0,
             # nt = [2,1]; w = [ln(3/3)=0, ln(3/2)]; c1 = [1]*0 -> zeros; c2: [.4*0, .6*ln1.5] -> [0,1]
             "dense": [[0.0, 0.0], [0.0, 1.0]]}
    ka = {
        "jaccard": [
            {"name": "toy5x3", "mat": toy, "u": toy_u},
            {"name": "selfloop", "mat": selfloop, "u": [[2, 2], [2, 2], [2, 2]]},
            {"name": "disjoint", "mat": disjoint, "u": disjoint_u},
            {"name": "duplicates", "mat": dups, "u": dups_u},
        ],
        "gficf": {"basic": gf, "filter": gf_filter, "w_zero_cell": gf_w0},
    }
    with open(os.path.join(OUT, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1)
    return ka


def jaccard_cases():
    out = {}
    cases = [(64, 5, 42), (1000, 15, 1), (1000, 30, 2), (3000, 15, 3), (1200, 50, 4), (700, 33, 5),
             (300, 100, 6), (400, 16, 7)]
    names = []
    for N, k, seed in cases:
        mat = synth.knn_windowed(N, k, seed=seed, perm_seed=seed + 100)
        rm, u = oracle.jaccard(mat, nthreads=4)
        assert np.array_equal(u, oracle_np.jaccard_counts_np(mat))
        assert np.array_equal(rm, oracle_np.jaccard_rmat(mat, u))
        nm = f"win_N{N}_k{k}"
        names.append(nm)
        out[nm + "_u"] = u.astype(np.uint8)
        out[nm + "_meta"] = np.array([N, k, seed], dtype=np.int64)
        out[nm + "_sha"] = np.frombuffer(sha(np.asfortranarray(rm)).encode(), dtype=np.uint8)
    # uniform random (u mostly 0) and a matrix with duplicate + self ids
    mat = synth.knn_uniform(2000, 20, seed=9)
    rm, u = oracle.jaccard(mat, nthreads=4)
    out["uniform_mat"] = mat
    out["uniform_u"] = u.astype(np.uint8)
    out["uniform_sha"] = np.frombuffer(sha(np.asfortranarray(rm)).encode(), dtype=np.uint8)
    r = synth.rand_u64(11, np.arange(500 * 12)).reshape(500, 12)
    mat = (r % np.uint64(40)).astype(np.int32) + 1        # ids in 1..40 of 500 cells: many duplicates / self ids
    rm, u = oracle.jaccard(mat, nthreads=4)
    assert np.array_equal(u, oracle_np.jaccard_counts_np(mat))
    out["dupheavy_mat"] = mat
    out["dupheavy_u"] = u.astype(np.uint8)
    out["dupheavy_sha"] = np.frombuffer(sha(np.asfortranarray(rm)).encode(), dtype=np.uint8)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "jaccard_cases.npz"), **out)


def gficf_cases():
    import scipy.sparse as sp

    out = {}
    for nm, (G, N, mn, mx, seed) in {"g600_n400": (600, 400, 0.05, 1.0, 7), "g300_n500_nofilter": (300, 500, 0.0, 1.0, 8),
                                     "g500_n300_max": (500, 300, 0.02, 0.6, 9)}.items():
        cp, ri, x = synth.counts_csc(G, N, seed=seed)
        r = oracle.gficf_csc(G, N, cp, ri, x, mn, mx)
        r2 = oracle_np.gficf_np(sp.csc_matrix((x, ri, cp), shape=(G, N)), mn, mx)
        assert np.array_equal(r["keep"], r2["keep"]) and np.allclose(r["x"], r2["gficf"].data, rtol=1e-13, atol=1e-15)
        out[nm + "_meta"] = np.array([G, N, seed], dtype=np.int64)
        out[nm + "_prop"] = np.array([mn, mx])
        out[nm + "_keep"] = r["keep"]
        out[nm + "_nt"] = r["nt"]
        out[nm + "_w"] = r["w"]
        out[nm + "_colptr"] = r["colptr"]
        out[nm + "_rowidx"] = r["rowidx"]
        out[nm + "_x"] = r["x"]
    np.savez_compressed(os.path.join(OUT, "gficf_cases.npz"), **out)


if __name__ == "__main__":
    known_answers()
    jaccard_cases()
    gficf_cases()
    print("golden fixtures written to", OUT)
