import time, numpy as np, torch
n = 454 * 2**20 // 8
src = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
torch.cuda.synchronize()
rt = torch.cuda.cudart()
for rep in range(3):
    dst = np.zeros(n)                      # fresh pageable destination, like an R vector
    t0 = time.perf_counter()
    torch.from_numpy(dst).copy_(src); torch.cuda.synchronize()
    t1 = time.perf_counter()
    dst2 = np.zeros(n)
    t2 = time.perf_counter()
    r = rt.cudaHostRegister(dst2.ctypes.data, dst2.nbytes, 0)
    t3 = time.perf_counter()
    torch.from_numpy(dst2).copy_(src, non_blocking=True); torch.cuda.synchronize()
    t4 = time.perf_counter()
    rt.cudaHostUnregister(dst2.ctypes.data)
    t5 = time.perf_counter()
    print(f"pageable D2H {1e3*(t1-t0):.1f} ms; register {1e3*(t3-t2):.1f} ms (rc {r}), copy {1e3*(t4-t3):.1f} ms, unregister {1e3*(t5-t4):.1f} ms")
