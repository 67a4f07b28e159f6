#!/usr/bin/env python3
"""gficf_jaccard_host (what `.Call("_gficf_rcpp_parallel_jaccard_coef")` binds): the 24 B/edge matrix copied back over PCIe against the
compact return (uint16 counts over PCIe into pinned staging + the reference's rows written by the host cores), ms per call on pageable
host buffers, a FRESH result buffer per call (as R allocates one), each setting in its own process (the switch is read once).
Usage: python tools/host_compact_ab.py | python tools/host_compact_ab.py child N k"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(N, k):
    import ctypes

    import numpy as np

    import gficf_amd
    import oracle
    from gficf_amd import _lib, synth

    L = _lib.load()
    mat = np.asfortranarray(synth.knn_windowed(N, k, seed=42, perm_seed=43))
    ctx = gficf_amd.default_context(0)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ts = []
    libc = ctypes.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes, libc.free.argtypes = ctypes.c_void_p, [ctypes.c_size_t], [ctypes.c_void_p]
    r_like = bool(os.environ.get("AB_MALLOC"))                     # a result buffer from plain malloc, as R's allocator hands one over (numpy asks
    for rep in range(6):                                           # for huge pages itself; R does not)
        nbytes = 24 * N * k
        if r_like:
            ptr = libc.malloc(nbytes)
            out = np.ctypeslib.as_array((ctypes.c_double * (3 * N * k)).from_address(ptr)).reshape(3, N * k)
        else:
            out = np.empty((3, N * k), dtype=np.float64)          # fresh pages every call
        t0 = time.perf_counter()
        rc = L.gficf_jaccard_host(ctx.handle, vp(mat), 0, N, k, N, ctypes.c_void_p(out.ctypes.data), 0)
        ts.append(time.perf_counter() - t0)
        assert rc == 0, _lib.last_error()
        if r_like and rep < 5:
            del out
            libc.free(ptr)
    cells = min(N, 512)
    want, _ = oracle.jaccard_cells(np.ascontiguousarray(mat), 0, cells, nthreads=os.cpu_count() or 1)
    ok = bool(np.array_equal(out[:, :cells * k].T, want))
    print(f"{min(ts[1:]) * 1e3:.4f} {sorted(ts[1:])[len(ts[1:]) // 2] * 1e3:.4f} {ok}")


def main():
    print(f"{'N':>8} {'k':>3} {'edges':>10} | {'full matrix over PCIe: best / median ms':>40} | {'compact return: best / median ms':>34} | compact / full (median)")
    shapes = ((3000, 15), (10000, 30), (54000, 30), (100000, 30), (100000, 50), (1000000, 30)) if not os.environ.get("AB_MALLOC") else ((10000, 30), (30000, 30), (54000, 30), (100000, 30), (1000000, 30))
    for N, k in shapes:
        res = []
        for lim in (str(1 << 62), "0"):
            env = dict(os.environ, GFICF_JACCARD_HOST_COMPACT_MIN_EDGES=lim)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(N), str(k)], capture_output=True, text=True, env=env, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.strip()]
            if r.returncode != 0 or not line:
                res.append((float("nan"), float("nan"), "FAILED " + r.stderr[-200:]))
            else:
                a, b, ok = line[-1].split()
                res.append((float(a), float(b), ok))
        (fa, fb, fo), (ca, cb, co) = res
        print(f"{N:>8} {k:>3} {N * k:>10} | {fa:>19.3f} / {fb:>17.3f} | {ca:>15.3f} / {cb:>15.3f} | {cb / fb:6.2f} x   {fo} {co}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
