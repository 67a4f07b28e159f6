#!/usr/bin/env python3
"""Does PCIe run full duplex for PAGEABLE host buffers when the two directions are driven from two host threads (each pageable copy holds
its thread)?  700 MB up + 700 MB down: one after the other against both at once; the same with pinned buffers.  Premise check for a
chunked upload / scale / download pipeline inside the GF-ICF host entry (round 5 lab; result in profiles/r05_duplex_probe.txt)."""
import threading
import time

import torch

n = 700 * 1024 * 1024 // 8
dev_a = torch.zeros(n, dtype=torch.float64, device="cuda")
dev_b = torch.ones(n, dtype=torch.float64, device="cuda")
for pinned in (False, True):
    h_up = torch.ones(n, dtype=torch.float64)
    h_dn = torch.empty(n, dtype=torch.float64)
    h_dn.zero_()                                     # pages mapped
    if pinned:
        h_up, h_dn = h_up.pin_memory(), h_dn.pin_memory()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def up():
        with torch.cuda.stream(s1):
            dev_a.copy_(h_up, non_blocking=True)
            s1.synchronize()

    def down():
        with torch.cuda.stream(s2):
            h_dn.copy_(dev_b, non_blocking=True)
            s2.synchronize()

    up(); down()
    for rep in range(3):
        t0 = time.perf_counter(); up(); t_up = time.perf_counter() - t0
        t0 = time.perf_counter(); down(); t_dn = time.perf_counter() - t0
        t0 = time.perf_counter()
        a, b = threading.Thread(target=up), threading.Thread(target=down)
        a.start(); b.start(); a.join(); b.join()
        t_both = time.perf_counter() - t0
        print(f"{'pinned' if pinned else 'pageable'}: up {t_up * 1e3:6.1f} ms ({0.734 / t_up:5.1f} GB/s)  down {t_dn * 1e3:6.1f} ms ({0.734 / t_dn:5.1f} GB/s)  "
              f"one after the other {1e3 * (t_up + t_dn):6.1f} ms  both at once from two threads {t_both * 1e3:6.1f} ms", flush=True)
