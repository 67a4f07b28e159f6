import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, scipy.sparse as sp
import gficf_amd
from gficf_amd import synth, _lib
from gficf_amd.api import _np_ptr, check, default_context
G, N = 23000, 54000
cp, ri, x = synth.counts_csc(G, N, seed=7)
L = _lib.load(); ctx = default_context()
cp64 = cp.astype(np.int64)
for rep in range(3):
    gk, nk = ctypes.c_int64(0), ctypes.c_int64(0)
    t0 = time.perf_counter()
    check(L.gficf_normalize_csc_host_plan(ctx.handle, G, N, _np_ptr(cp64), 1, _np_ptr(ri), _np_ptr(x), 0.05, 1.0, None, ctypes.byref(gk), ctypes.byref(nk)))
    t1 = time.perf_counter()
    keep = np.zeros(G, np.uint8); nt = np.zeros(G, np.int64); w = np.zeros(G); ocp = np.zeros(N + 1, np.int64)
    ori = np.empty(nk.value, np.int32); ox = np.empty(nk.value)
    t2 = time.perf_counter()
    check(L.gficf_normalize_csc_host_finish(ctx.handle, _np_ptr(keep), _np_ptr(nt), _np_ptr(w), _np_ptr(ocp), _np_ptr(ori), _np_ptr(ox)))
    t3 = time.perf_counter()
    print(f"plan {1e3*(t1-t0):.1f} ms (H2D {(len(ri)*12)/1e6:.0f} MB), alloc outputs {1e3*(t2-t1):.1f} ms, finish {1e3*(t3-t2):.1f} ms (D2H {nk.value*12/1e6:.0f} MB)")
M = sp.csc_matrix((x, ri, cp), shape=(G, N))
t0 = time.perf_counter(); r = gficf_amd.gficf(M, normalize=False, verbose=False); print(f"gficf() total {1e3*(time.perf_counter()-t0):.1f} ms")
