"""One fresh process = one sample of the epoch-time distribution: create the bench-size handle, run a few
epochs, print the kernel time.  No torch (fast start), the matrix is cached under /tmp on the GPU box between processes.
   python3 tools/r02/mode_probe.py [tag] [epochs] [recreate]     (under rocprofv3 --pmc ... -- python3 ...)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
import geglove                          # noqa: E402
from geglove import synth               # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "probe"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
recreate = int(sys.argv[3]) if len(sys.argv) > 3 else 1
V, D, nnz = 625_000, 200, 125_000_000
cache = "/tmp/ge_mode_probe_%d_%d.npz" % (V, nnz)
if os.path.exists(cache):
    z = np.load(cache); I, J, X, xmax = z["I"], z["J"], z["X"], float(z["xmax"])
else:
    I, J, X, xmax = synth.synthetic_coo_shard(V, (0, V), nnz, seed=0xC0FFEE)
    np.savez(cache, I=I, J=J, X=X, xmax=xmax)
out = {"tag": tag, "pid": os.getpid(), "runs": []}
for r in range(recreate):
    cfg = geglove.Configuration({"graph": "synthetic", "method": "glove", "dim": D, "threads": 1,
                                 "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
                                 "opt": {"method": "adagrad", "tolerance": 0, "maxiter": epochs}, "output": {"uri": []},
                                 "device": {"mode": "hogwild", "shuffle": "device", "seed": 42}})
    t0 = time.perf_counter()
    opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
    tc = time.perf_counter() - t0
    ms = []
    for it in range(epochs):
        opt.epoch(it); ms.append(round(opt.last_kernel_ms()[0], 3))
    run = {"create_s": round(tc, 2), "kernel_ms": ms}
    if os.environ.get("GE_PROBE_CHASE"):
        import ctypes as C
        ch = C.CDLL(os.path.join(ROOT, "tools", "micro", "libchase.so"))
        ch.chase.restype = C.c_double
        ch.chase.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int]
        for name in ("context", "focus", "gsq_context", "gsq_focus"):
            ptr, n = opt.device_ptr(name)
            run["ptr_" + name] = hex(ptr)
            run["chase_idle_" + name] = round(ch.chase(ptr, V, 816, 20000, 1), 1)            # one wave: latency of a random row, TLB miss included
            run["chase_small_" + name] = round(ch.chase(ptr, 20000, 816, 20000, 1), 1)      # 16 MB: translations cached
            run["chase_load_" + name] = round(ch.chase(ptr, V, 816, 4000, 4096), 1)         # 4096 waves at once
    if os.environ.get("GE_PROBE_XCD"):
        import ctypes as C
        ch = C.CDLL(os.path.join(ROOT, "tools", "micro", "libchase.so"))
        ch.xcd_stream.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        hipl = C.CDLL(None)
        buf = C.c_void_p()
        hipl.hipMalloc(C.byref(buf), C.c_size_t(256 * (16 << 20)))
        hipl.hipMemset(buf, 1, C.c_size_t(256 * (16 << 20)))
        g = (C.c_double * 8)(); mm = (C.c_double * 3)()
        ch.xcd_stream(buf, 256, 1, g, mm)
        ch.xcd_stream(buf, 256, 4, g, mm)
        run["xcd_GBps"] = [round(x, 1) for x in g]; run["xcd_total_TBps"] = round(256 * 4 * (16 << 20) / (mm[2] * 1e-3) / 1e12, 3)
        run["xcd_ticks_min_max"] = [mm[0], mm[1]]
        hipl.hipFree(buf)
    out["runs"].append(run)
    opt.close()
print(json.dumps(out), flush=True)
