#!/usr/bin/env python3
"""What pinned host memory would buy the builder's copy-out (520 MB of J and X): the cost of pinning against the copy rates."""
import time, torch
n = 130_000_000                       # floats: 520 MB
d = torch.empty(n, dtype=torch.float32, device="cuda"); d.fill_(1.0); torch.cuda.synchronize()
for rep in range(3):
    t = time.perf_counter(); h = torch.empty(n, dtype=torch.float32, pin_memory=True); t_pin = time.perf_counter() - t
    t = time.perf_counter(); h.copy_(d); torch.cuda.synchronize(); t_copy = time.perf_counter() - t
    t = time.perf_counter(); d.copy_(h); torch.cuda.synchronize(); t_up = time.perf_counter() - t
    t = time.perf_counter(); p = torch.empty(n, dtype=torch.float32); t_alloc = time.perf_counter() - t
    t = time.perf_counter(); p.copy_(d); torch.cuda.synchronize(); t_pcopy = time.perf_counter() - t
    t = time.perf_counter(); p.copy_(d); torch.cuda.synchronize(); t_pcopy2 = time.perf_counter() - t
    t = time.perf_counter(); d.copy_(p); torch.cuda.synchronize(); t_pup = time.perf_counter() - t
    print("520 MB: pin %.1f ms, D2H into pinned %.1f ms (%.1f GB/s), H2D from pinned %.1f ms; pageable: alloc %.1f ms, first D2H %.1f ms, second D2H %.1f ms, H2D %.1f ms"
          % (t_pin * 1e3, t_copy * 1e3, 0.52 / t_copy, t_up * 1e3, t_alloc * 1e3, t_pcopy * 1e3, t_pcopy2 * 1e3, t_pup * 1e3), flush=True)
    del h, p
