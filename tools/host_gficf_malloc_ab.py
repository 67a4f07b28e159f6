#!/usr/bin/env python3
"""gficf_normalize_csc_host_plan + _finish (what `.Call("_gficf_gficf_csc")` binds) at the config 3 shape with the big result vectors
(@i, @x of the kept entries) from plain malloc, as R's allocator hands them over, without and with the library's request for transparent
huge pages on them (gficf_prefault -> gficf_advise_hugepages; GFICF_HIP_NO_HUGEPAGE=1 turns it off).  Each setting in its own process.
Usage: python tools/host_gficf_malloc_ab.py | ... child"""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import torch

    import bench
    import gficf_amd
    from gficf_amd import _lib

    G, N = bench.GFICF_G, bench.GFICF_N
    colptr, rowidx, x = bench.synth_counts_device(torch, G, N)
    cp, ri, xv = colptr.cpu().numpy().astype(np.int32), rowidx.cpu().numpy(), x.cpu().numpy()
    del colptr, rowidx, x
    torch.cuda.empty_cache()
    L = _lib.load()
    ctx = gficf_amd.default_context(0)
    libc = ctypes.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes, libc.free.argtypes = ctypes.c_void_p, [ctypes.c_size_t], [ctypes.c_void_p]
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ts = []
    for rep in range(4):
        gk, nk = ctypes.c_int64(0), ctypes.c_int64(0)
        t0 = time.perf_counter()
        rc = L.gficf_normalize_csc_host_plan(ctx.handle, G, N, vp(cp), 0, vp(ri), vp(xv), ctypes.c_double(0.05), ctypes.c_double(1.0), None, ctypes.byref(gk), ctypes.byref(nk))
        assert rc == 0, _lib.last_error()
        t1 = time.perf_counter()
        keep, nt, w, ocp = np.empty(G, np.uint8), np.empty(G, np.int64), np.empty(G, np.float64), np.empty(N + 1, np.int32)
        p_i, p_x = libc.malloc(4 * nk.value), libc.malloc(8 * nk.value)
        rc = L.gficf_normalize_csc_host_finish(ctx.handle, vp(keep), vp(nt), vp(w), vp(ocp), ctypes.c_void_p(p_i), ctypes.c_void_p(p_x))
        assert rc == 0, _lib.last_error()
        t2 = time.perf_counter()
        ts.append((t2 - t0, t1 - t0, t2 - t1))
        s = float(np.ctypeslib.as_array((ctypes.c_double * 16).from_address(p_x)).sum())
        libc.free(p_i); libc.free(p_x)
    best = min(ts[1:])
    print(f"{best[0] * 1e3:.2f} {best[1] * 1e3:.2f} {best[2] * 1e3:.2f} {nk.value} {s:.6f}")


def main():
    print("config 3 shape (23 k genes x 54 k cells, 88.5 M stored entries), result vectors from malloc: ms per call = plan + finish")
    for label, env in (("library madvise(MADV_HUGEPAGE) off", {"GFICF_HIP_NO_HUGEPAGE": "1"}), ("library madvise(MADV_HUGEPAGE) on (default)", {})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=400)
        line = [l for l in r.stdout.splitlines() if l.strip()]
        if r.returncode != 0 or not line:
            print(f"{label}: FAILED {r.stderr[-300:]}")
            continue
        a, b, c, nk, s = line[-1].split()
        print(f"{label:46s}: {a:>7} ms = plan {b:>6} + finish {c:>6}   (kept entries {nk})", flush=True)


if __name__ == "__main__":
    child() if len(sys.argv) > 1 and sys.argv[1] == "child" else main()
