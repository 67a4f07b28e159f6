import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gficf_amd
ops = gficf_amd.HipOps(0)
N, k, nloc = 800000, 30, 100000
kp, pw = ops.kpad(k), ops.packed_words(N, k)
table = torch.randint(1, N, (N, kp), dtype=torch.int32, device="cuda")
packed = torch.empty((N, pw), dtype=torch.int32, device="cuda")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("packed words", pw, "bytes/row", pw * 4)
print("pack 100k rows: %.1f us" % t(lambda: ops.jaccard_pack_rows(table[:nloc], nloc, k, N, packed[:nloc])))
print("unpack 700k rows: %.1f us" % t(lambda: ops.jaccard_unpack_rows(packed[nloc:], N - nloc, k, N, table[nloc:])))
