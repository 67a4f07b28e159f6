"""The GF-ICF host entry at the config-3 shape with and without the values of M[keep, ] ($rawCounts): gficf_normalize_csc_host_plan + _finish against
plan + _finish_raw (host threads gather the kept x from the caller's vectors while the result crosses PCIe), result vectors fresh per call (as R and the
mirror allocate them) and reused; and the gather alone (gficf_csc_kept_values_host).  ms per call, median of 5."""
import ctypes
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import gficf_amd
from gficf_amd import _lib, synth
from gficf_amd.api import _np_ptr

G, N = 23000, 54000
cp, ri, x = synth.counts_csc(G, N, seed=7)
ri = ri.astype(np.int32)
L, ctx = _lib.load(), gficf_amd.default_context()
gk, nk = ctypes.c_int64(0), ctypes.c_int64(0)


def call(raw, bufs=None, own_ids=True):
    t0 = time.perf_counter()
    assert L.gficf_normalize_csc_host_plan(ctx.handle, G, N, _np_ptr(cp), 1, _np_ptr(ri), _np_ptr(x), 0.05, 1.0, None, ctypes.byref(gk), ctypes.byref(nk)) == 0
    t1 = time.perf_counter()
    n = nk.value
    b = bufs or dict(keep=np.zeros(G, np.uint8), ocp=np.zeros(N + 1, np.int64), ori=np.empty(n, np.int32), ox=np.empty(n), rri=np.empty(n, np.int32), rx=np.empty(n))
    if raw:
        rc = L.gficf_normalize_csc_host_finish_raw(ctx.handle, _np_ptr(b["keep"]), None, None, _np_ptr(b["ocp"]), _np_ptr(b["ori"]), _np_ptr(b["ox"]), _np_ptr(ri), _np_ptr(x),
                                                   _np_ptr(b["rri"]) if own_ids else None, _np_ptr(b["rx"]))
    else:
        rc = L.gficf_normalize_csc_host_finish(ctx.handle, _np_ptr(b["keep"]), None, None, _np_ptr(b["ocp"]), _np_ptr(b["ori"]), _np_ptr(b["ox"]))
    assert rc == 0
    t2 = time.perf_counter()
    return 1e3 * (t1 - t0), 1e3 * (t2 - t1), b


call(True)
print(f"config 3 shape: {G} genes x {N} cells, {len(x)} stored entries, {nk.value} kept")
for label, raw, reuse, own in (("finish, fresh result vectors", False, False, True), ("finish_raw (values + row ids), fresh result vectors", True, False, True),
                               ("finish_raw (values only: @i shared), fresh result vectors", True, False, False),
                               ("finish, reused result vectors", False, True, True), ("finish_raw (values + row ids), reused result vectors", True, True, True),
                               ("finish_raw (values only), reused result vectors", True, True, False)):
    bufs = call(raw)[2] if reuse else None
    t = [call(raw, bufs, own)[:2] for _ in range(5)]
    print(f"{label:62s}: plan {statistics.median(a for a, _ in t):6.2f} + finish {statistics.median(b for _, b in t):6.2f} ms   (finish calls: {' '.join(f'{b:.1f}' for _, b in t)})")
b = call(True)[2]
keep, ocp = b["keep"], b["ocp"]
for own in (True, False):
    t = []
    for _ in range(5):
        rri, rx = np.empty(nk.value, np.int32), np.empty(nk.value)
        t0 = time.perf_counter()
        assert L.gficf_csc_kept_values_host(G, N, _np_ptr(cp), 1, _np_ptr(ri), _np_ptr(x), _np_ptr(keep), _np_ptr(ocp), _np_ptr(rri) if own else None, _np_ptr(rx)) == 0
        t.append(1e3 * (time.perf_counter() - t0))
    print(f"the gather alone, fresh vectors, {'values + row ids' if own else 'values only':17s}: {statistics.median(t):6.2f} ms   ({' '.join(f'{v:.1f}' for v in t)})")
assert np.array_equal(rx, x[keep[ri].astype(bool)])
print(f"host threads available: {os.cpu_count()}")
