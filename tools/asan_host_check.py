"""Host-side sanitizer pass over the library's pure-host entries (no GPU, no torch): random shapes through the AddressSanitizer + UBSan build
(`make -C gficf_amd/csrc asan` -> build_asan/libgficf_hip_asan.so), results compared with numpy / scipy.

    LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python tools/asan_host_check.py [seconds]

Entries: gficf_csc_kept_values_host (the values of M[keep, ]; tails of cells that keep nothing, empty cells, nothing kept, both pointer widths),
gficf_jaccard_expand_host (compact return -> the reference's (N k) x 3 matrix; int32 and double ids, ld > N), the format queries.
(The first version of the kept-values gather wrote one slot past the vector — found by the GPU fuzz run as a heap corruption; this pass finds
that class of bug in seconds and without a GPU.)"""
import ctypes
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gficf_amd", "csrc", "build_asan", "libgficf_hip_asan.so")
L = ctypes.CDLL(LIB)
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
L.gficf_csc_kept_values_host.argtypes = [i64, i64, vp, i32, vp, vp, vp, vp, vp, vp]
L.gficf_jaccard_expand_host.argtypes = [vp, i32, i64, i32, i64, vp, vp, i32]
L.gficf_jaccard_kpad.argtypes = [i32]
L.gficf_jaccard_row_words.argtypes = [i64, i32]
L.gficf_jaccard_packed_words.argtypes = [i64, i32]
L.gficf_last_error.restype = ctypes.c_char_p
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0, n_kv, n_ex = time.time(), 0, 0


def ptr(a):
    return a.ctypes.data if a is not None and a.size else (a.ctypes.data if a is not None else None)


while time.time() - t0 < budget:
    # ---- the values of M[keep, ]
    G, N = int(rng.integers(1, 400)), int(rng.integers(0, 300))
    big = rng.random() < 0.05
    if big:
        G, N = int(rng.integers(2000, 4000)), int(rng.integers(4000, 9000))           # several host threads
    M = sp.random(G, N, density=float(rng.choice([0.0, 0.02, 0.2, 0.6])) if not big else 0.25, format="csc", random_state=rng, dtype=np.float64)
    M.data = np.ceil(M.data * 7)
    if M.nnz:
        M.data[:: int(rng.integers(2, 50))] = 0.0
    keep = (rng.random(G) < float(rng.choice([0.0, 0.1, 0.7, 1.0]))).astype(np.uint8)
    if N and rng.random() < 0.5:                              # a tail (and a head) of cells that keep nothing
        M = M.tolil() if not big else M
        if not big:
            drop = np.flatnonzero(keep == 0)
            for c in list(range(max(0, N - int(rng.integers(1, 4))), N)) + [0]:
                M[:, c] = 0
                if len(drop):
                    M[drop[:2], c] = 3.0
            M = M.tocsc()
    pt = np.int64 if rng.random() < 0.5 else np.int32
    want = M[np.flatnonzero(keep), :]
    cp, ri, x = M.indptr.astype(pt), M.indices.astype(np.int32), np.ascontiguousarray(M.data, dtype=np.float64)
    kcp = want.indptr.astype(pt)
    with_ids = rng.random() < 0.5
    oi = np.full(want.nnz, -7, dtype=np.int32)               # exactly sized: a store one slot past is a heap overflow ASan reports
    ox = np.full(want.nnz, np.nan)
    rc = L.gficf_csc_kept_values_host(G, N, cp.ctypes.data, int(pt is np.int64), ri.ctypes.data, x.ctypes.data, keep.ctypes.data, kcp.ctypes.data,
                                      oi.ctypes.data if with_ids else None, ox.ctypes.data)
    assert rc == 0, L.gficf_last_error()
    assert np.array_equal(ox, want.data) and (not with_ids or np.array_equal(oi, want.indices)), (G, N)
    if want.nnz and N > 1 and rng.random() < 0.5:           # a column pointer that is not M[keep, ]'s: refused, and nothing is stored outside the vector
        bad = kcp.copy()
        c = int(rng.integers(1, N + 1))
        bad[c:] += pt(rng.choice([-1, 1]))
        bad = np.maximum.accumulate(np.clip(bad, 0, None))
        if not np.array_equal(bad, kcp):
            ox2 = np.full(max(int(bad[-1]), 0), np.nan)
            rc = L.gficf_csc_kept_values_host(G, N, cp.ctypes.data, int(pt is np.int64), ri.ctypes.data, x.ctypes.data, keep.ctypes.data, bad.ctypes.data, None, ox2.ctypes.data)
            assert rc != 0
    n_kv += 1
    # ---- compact Jaccard return -> the (N k) x 3 matrix
    N, k = int(rng.integers(1, 3000)), int(rng.integers(1, 70))
    if rng.random() < 0.03:
        N, k = int(rng.integers(40000, 90000)), 30               # above the threaded expansion's threshold
    ld = N + int(rng.integers(0, 5))
    f64 = rng.random() < 0.5
    idx = rng.integers(1, N + 1, size=(k, ld)).astype(np.float64 if f64 else np.int32)       # column-major N x k with pitch ld
    u = rng.integers(0, k + 1, size=N * k).astype(np.uint16)
    out = np.full((3, N * k), np.nan)                          # column-major (N k) x 3
    rc = L.gficf_jaccard_expand_host(idx.ctypes.data, int(f64), N, k, ld, u.ctypes.data, out.ctypes.data, int(rng.integers(0, 9)))
    assert rc == 0, L.gficf_last_error()
    uu = u.astype(np.float64)
    pos = uu > 0
    src = np.repeat(np.arange(1, N + 1, dtype=np.float64), k)
    dst = idx[:, :N].T.reshape(-1).astype(np.float64)
    assert np.array_equal(out[0], np.where(pos, src, 0.0)) and np.array_equal(out[1], np.where(pos, dst, 0.0))
    assert np.array_equal(out[2], np.where(pos, uu / (2.0 * k - uu), 0.0))
    n_ex += 1
for k in list(range(0, 300)) + [513, 65535, 65536, -1]:
    L.gficf_jaccard_kpad(k)
    for n in (0, 1, 131070, 131071, 10**6, 2**31 - 1, 2**31):
        L.gficf_jaccard_row_words(n, k)
        L.gficf_jaccard_packed_words(n, k)
print(f"tools/asan_host_check.py: {n_kv} kept-values cases and {n_ex} expansions in {time.time() - t0:.0f} s under AddressSanitizer + UBSan: no report, every result equal to numpy / scipy")
