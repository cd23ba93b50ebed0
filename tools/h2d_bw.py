"""Aggregate host-to-device rate of wire-batch-sized copies (3.8 MB, pinned) from T threads on their own streams: python tools/h2d_bw.py"""
import torch, time, threading
n_bytes = 3_800_000
for T in (1, 4, 8, 16, 24):
    hs = [torch.empty(n_bytes, dtype=torch.uint8).pin_memory() for _ in range(T)]
    ds = [torch.empty(n_bytes, dtype=torch.uint8, device="cuda") for _ in range(T)]
    ss = [torch.cuda.Stream() for _ in range(T)]
    R = 200
    def work(i):
        with torch.cuda.stream(ss[i]):
            for _ in range(R):
                ds[i].copy_(hs[i], non_blocking=True)
                ss[i].synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"T={T}: {T*R*n_bytes/dt/1e9:.1f} GB/s aggregate, {1e3*dt/R:.3f} ms per copy round")
