#!/usr/bin/env python3
"""bench.py -- GloVe pair-updates/sec at dim=200 on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (ge_glove_epoch: the AdaGrad pair-update kernel over every
nonzero this rank owns) over the synthetic co-occurrence matrix; inputs are resident in HBM before
the timed region.  Weak scaling: every GPU owns ROWS_PER_GPU focus rows and ~NNZ_PER_GPU nonzeros,
the context factors are replicated and reconciled by an all-reduce of the per-rank deltas (RCCL) per step,
so at 8 GPUs the job is BASELINE config C4 (5 M vertices / 1 B nonzeros / dim 200).

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and, at N=1,
`cpu_baseline` (the oracle's restatement of Adagrad.createJob on the host cores, T = cores - 1 and T = 1).

roofline: `achieved` = bytes the kernel's schedule has to move per launch (ge_glove_info.schedule_bytes:
per nonzero 20 B of matrix + the streamed row pair loaded and stored, per run the resident row pair loaded and
published) / the kernel's HIP-event time; `frac` = achieved / 8 TB/s, never above 1.  The SURVEY 8(d) figure
(16*D+28 read + 16*D+16 written per update, as if no row stayed in registers) is reported beside it under
`naive_schedule` -- it describes a schedule this kernel does not run.  `traffic` = HBM bytes per launch from the
committed rocprofv3 counter passes of the same kernel and workload (profiles/traffic.json), else null.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "graph-embeddings_amd"))

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dim", type=int, default=200)
    ap.add_argument("--rows-per-gpu", type=int, default=625_000)
    ap.add_argument("--nnz-per-gpu", type=int, default=125_000_000)
    ap.add_argument("--method", default="glove", choices=["glove", "pglove"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--hot", default="auto", choices=["auto", "none", "all"])
    ap.add_argument("--hot-theta", type=float, default=0.0, help="column j is a hub when count(j) >= theta * N / workers (0 = the library default, 0.25)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="storage of the embedding rows; bf16 = BASELINE config C5 (fp32 accumulators, stochastic rounding)")
    ap.add_argument("--opt", default="adagrad", choices=["adagrad", "adam", "amsgrad"],
                    help="update rule (the headline metric is quoted on adagrad; adam/amsgrad keep two moment rows per side)")
    ap.add_argument("--sync-every", type=int, default=1, help="steps between context all-reduces (N>1)")
    ap.add_argument("--wire", default="bf16", choices=["bf16", "f32"], help="dtype of the deltas on the wire (N>1)")
    ap.add_argument("--exchange", default="overlap", choices=["overlap", "sync"],
                    help="N>1: 'overlap' runs the all-reduce of step k's context deltas under step k+1 (they land one step late); "
                         "'sync' exchanges after every step before the next one starts")
    ap.add_argument("--sync-first", type=int, default=2, help="N>1 with --exchange overlap: this many first steps are exchanged synchronously (at most --warmup of them)")
    ap.add_argument("--reserve-waves", type=int, default=256,
                    help="N>1 with --exchange overlap: wavefront slots the epoch kernel leaves free for the all-reduce kernels")
    ap.add_argument("--workers", type=int, default=0, help="sequential workers (wavefronts); 0 = fill the device (cfg.workers)")
    ap.add_argument("--layout", default="", help="comma list of ge_glove_cfg.layout_flags: fixed_cuts, plain_long_rows, separate_tables, packed_records, first_placement (default: none)")
    ap.add_argument("--shards-on-one-gpu", type=int, default=1,
                    help="N=1 only: K > 1 puts the matrix of a K-GPU job on ONE GPU (V = K x rows-per-gpu, the K ranks' shards concatenated: "
                         "8 = BASELINE C4 at its own size, 5 M vertices / 0.86 G nonzeros); a recorded profile leg, not the default workload")
    ap.add_argument("--hub-segments", type=int, default=0,
                    help="N>1: small exchanges of the hub rows per epoch (ge_sync_epoch); 0 = the library's choice (by the busiest column; at least max(8, ranks)), -1 = none (one exchange per epoch for every row)")
    ap.add_argument("--no-other-form", action="store_true", help="N>1: do not time the other exchange form behind the quoted region")
    ap.add_argument("--accum-sync-every", type=int, default=4, help="every how many context syncs the AdaGrad accumulators are reconciled too (N>1)")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, V, D, I, J, X, xmax):
    """Oracle (CPU restatement, kind 'port') timed on a bounded sample of the same workload: T = cores - 1 racing
    threads (the reference's default, Configuration.java:71-73) and T = 1.  Only this leg of bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle as O
    O.build()
    cores = max((os.cpu_count() or 2) - 1, 1)          # Configuration.getThreads() default: cores - 1
    kind = O.COST_GLOVE if args.method == "glove" else O.COST_PGLOVE
    n = len(I)
    order = np.random.default_rng(1).permutation(n)        # random order, like the shuffled epoch

    def run(m, threads, warm):
        sel = order[:m]
        g = O.Glove(V, D, I[sel], J[sel], X[sel], xmax, kind, seed=42, threads=threads)
        if warm:
            g.epoch(race=threads > 1, shuffle=False)       # touch the tables, start from warm caches/pages
        t0 = time.perf_counter()
        g.epoch(race=threads > 1, shuffle=False)
        dt = time.perf_counter() - t0
        g.close()
        return m / dt, dt

    def leg(threads, seconds):
        probe = min(1_000_000 if threads == 1 else 2_000_000, n)
        rate, _ = run(probe, threads, True)
        m = int(min(n, max(probe, rate * seconds)))
        rate, dt = run(m, threads, True)
        return rate, dt, m

    rate_t, dt_t, m_t = leg(cores, args.cpu_seconds * 0.6)
    rate_1, dt_1, m_1 = leg(1, args.cpu_seconds * 0.4)
    return {"value": rate_t, "unit": "pair-updates/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "single_thread": {"value": rate_1, "cores": 1, "sample_nonzeros": m_1, "seconds": dt_1},
            "sample": "%d nonzeros drawn at random from the same matrix (same V x D tables), 1 timed Hogwild pass with %d racing threads "
                      "after a warm-up pass, %.1f s; then %d nonzeros with one thread, %.1f s; "
                      "C restatement of Adagrad.createJob (flat arrays: faster than the Java loop, baseline only)"
                      % (m_t, cores, dt_t, m_1, dt_1)}


def launch_ranks(n):
    """`python3 bench.py --gpus N` from a plain shell: this process becomes the launcher.  It starts N copies of itself, one rank
    per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT as torch.distributed.run would set them), waits
    for them and exits with their status.  It never imports torch and never touches a GPU (nothing may exec or fork after a HIP
    call on this pool), and it never prints a bench line itself: rank 0 does, or nobody does and the exit code is non-zero."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", GE_BENCH_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    status = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                print("bench.py: rank %d exited with status %d; stopping the other ranks" % (procs.index(p), rc), file=sys.stderr, flush=True)
                for q in alive:                                     # exactly the processes started above, by handle
                    q.terminate()
                t_end = time.time() + 20
                for q in alive:
                    try:
                        q.wait(timeout=max(0.1, t_end - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
    raise SystemExit(status)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("GE_BENCH_LAUNCHED") != "1":
        launch_ranks(args.gpus)                                   # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: a bench line must carry the number of GPUs it ran on" % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import geglove
    from geglove import capi, synth, parallel

    dist = None
    if world > 1:
        import torch.distributed as dist
        # GE_BENCH_BACKEND=gloo + GE_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank flow with every rank on GPU 0
        # (RCCL refuses two ranks per device); never used by the driver.
        if os.environ.get("GE_BENCH_ONE_DEVICE") == "1":
            local_rank = 0
        if local_rank >= torch.cuda.device_count():             # counting devices does not initialise the GPU
            raise SystemExit("bench.py --gpus %d: rank %d has no GPU (%d visible); no line is printed for a run that cannot have %d GPUs"
                             % (world, rank, torch.cuda.device_count(), world))
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("GE_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if capi.lib().ge_device_count() <= 0:
        raise SystemExit("bench.py needs a gfx950 GPU: " + capi.lib().ge_last_error().decode())

    D = args.dim
    K = args.shards_on_one_gpu if world == 1 else 1
    V = args.rows_per_gpu * world * K
    rows = parallel.shard_rows(V, world, rank)
    t_gen = time.perf_counter()
    if K > 1:            # the whole K-GPU matrix on this GPU: every rank's shard from the generator that rank would run
        parts = [synth.synthetic_coo_shard(V, parallel.shard_rows(V, K, k), args.nnz_per_gpu, seed=0xC0FFEE) for k in range(K)]
        I, J, X = (np.concatenate([q[c] for q in parts]) for c in range(3))
        xmax = parts[0][3]
        del parts
    else:
        I, J, X, xmax = synth.synthetic_coo_shard(V, rows, args.nnz_per_gpu, seed=0xC0FFEE)
    t_gen = time.perf_counter() - t_gen
    n_local = int(I.shape[0])

    cfg = geglove.Configuration({
        "graph": "synthetic", "method": args.method, "dim": D, "threads": 1,
        "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
        "opt": {"method": args.opt, "tolerance": 0, "maxiter": args.steps},
        "output": {"uri": []},
        "device": {"mode": "hogwild", "shuffle": "device", "seed": 42, "id": local_rank, "hot": args.hot, **({"hot_theta": args.hot_theta} if args.hot_theta > 0 else {}), "dtype": args.dtype, "layout": [x for x in args.layout.split(",") if x],
                   "workers": args.workers if args.workers else (-args.reserve_waves if (world > 1 and args.exchange == "overlap") else 0),
                   "row_range": rows if world > 1 else (0, 0)}})
    t_create = time.perf_counter()
    opt = geglove.createOptimizer(cfg, geglove.CooMatrix(V, I, J, X, xmax))
    t_create = time.perf_counter() - t_create

    sync = None
    if world > 1 and args.opt != "adagrad":
        raise SystemExit("bench.py --gpus N: the context exchange (which deltas add, which average) is defined for adagrad only")
    if world > 1:
        sync = parallel.context_sync_for(opt, torch.device("cuda", local_rank), lazy_every=args.accum_sync_every, wire=args.wire)
        hub_plan = sync.hub_plan(max(args.hub_segments, 0))

    def step(it, form=None):
        # N > 1: ge_sync_epoch -- the rank's epoch in segments, the hub rows of the context side reconciled behind each
        c = sync.epoch(it, args.hub_segments) if (sync is not None and args.hub_segments >= 0) else opt.epoch(it)
        if sync is not None and (it + 1) % args.sync_every == 0:
            # the first epochs of a run move the model most: they are exchanged synchronously (no epoch-2 bump from landing everything one
            # step late), the overlapped form takes over after them -- never inside the timed region (capped by --warmup)
            if (form or args.exchange) == "overlap" and it >= min(args.sync_first, args.warmup):
                sync.turn()                         # lands the deltas sent one exchange ago and sends this step's: the all-reduce runs under the next epoch
            else:
                sync.sync()
        return c

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    costs = []
    for w in range(args.warmup):
        costs.append(step(w))
    fence()
    kernel_ms = 0.0
    launches = 0
    t0 = time.perf_counter()
    for k in range(args.steps):
        costs.append(step(args.warmup + k))
        ms, nl = opt.last_kernel_ms()
        kernel_ms += ms
        launches += nl
    fence()
    dt = time.perf_counter() - t0

    # N > 1: the OTHER exchange form on the same handles, timed the same way right behind the quoted region (the quoted `value`
    # is the form --exchange names; a scaling record then holds both: the overlapped form is faster, the synchronous one
    # trains the better model per epoch).  The handle keeps its worker count (reserved slots) in both.
    other = None
    if sync is not None and not args.no_other_form:
        form2 = "sync" if args.exchange == "overlap" else "overlap"
        costs2, k_ms2, nl2 = [], 0.0, 0
        step(args.warmup + args.steps, form2); fence()          # one untimed step to change over
        t2 = time.perf_counter()
        for k in range(args.steps):
            costs2.append(step(args.warmup + args.steps + 1 + k, form2))
            ms, nl = opt.last_kernel_ms()
            k_ms2 += ms; nl2 += nl
        fence()
        other = {"exchange": form2, "dt": time.perf_counter() - t2, "costs": costs2, "kernel_ms": k_ms2 / max(args.steps, 1)}

    # what THIS box's memory system gives a plain device-to-device copy (1 GiB, read + written bytes), right after the timed
    # region: boxes of the pool differ by 10-15 % in the epoch time of one binary, and this says which kind ran
    box_copy = None
    if rank == 0:
        import ctypes
        g = ctypes.c_double(0.0)
        if capi.lib().ge_copy_bandwidth(local_rank, 1 << 30, 5, ctypes.byref(g)) == 0:
            box_copy = g.value

    total_updates = n_local * args.steps
    n_global = n_local
    kernel_ms_max = kernel_ms / max(args.steps, 1)            # this rank's epoch kernel(s), own time, per step (a segmented epoch: its launches summed)
    if dist is not None:
        t = torch.tensor([dt, kernel_ms_max, other["dt"] if other else 0.0, other["kernel_ms"] if other else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kernel_ms_max = float(t[0].item()), float(t[1].item())
        if other:
            other["dt"], other["kernel_ms"] = float(t[2].item()), float(t[3].item())
        # the cost the ranks must agree on (Optimizer.java:96-97: the sum of the jobs' costs over the number of nonzeros), per step
        u = torch.tensor([float(total_updates), float(n_local)] + [float(c) for c in costs] + ([float(c) for c in other["costs"]] if other else []),
                         dtype=torch.float64, device="cuda")
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total_updates, n_global = float(u[0].item()), float(u[1].item())
        costs = [float(x) for x in u[2:2 + len(costs)].tolist()]
        if other:
            other["costs"] = [float(x) for x in u[2 + len(costs):].tolist()]

    if rank == 0:
        info = opt.info()
        kernel = "k_adagrad_runs<%d, %d, %s, %s, %s>" % (info["vector_width"], info["chunks_per_lane"], {"adagrad": 0, "adam": 1, "amsgrad": 2}[args.opt],
                                                          "true" if args.dtype == "bf16" else "false",
                                                          "true" if (D // info["vector_width"]) % 64 != 0 else "false")     # fat rows (bf16: fat accumulator rows)
        # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE doubled per the gfx950 correction +
        # WRITE_SIZE); an entry counts only for the kernel instance, workload and layout it was taken on
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath) and world == 1:
            try:
                for t in json.load(open(tpath)):
                    if (t.get("kernel"), t["vocab"], t["nnz_per_gpu"], t["dim"], t["cost"], t.get("layout", ""), t.get("schedule_bytes"), t.get("row_stride")) == \
                            (kernel, V, n_local, D, args.method, args.layout, info["schedule_bytes"], info["row_stride"]):
                        traffic = t["traffic_bytes_per_launch"]
            except Exception:
                traffic = None
        # SURVEY.md 8d bytes per pair-update (8f for the moment optimisers: one more row per side): the naive schedule
        read_b, write_b = (16 * D + 28, 16 * D + 16) if args.opt == "adagrad" else (24 * D + 36, 24 * D + 24)
        if args.dtype == "bf16":
            read_b, write_b = 12 * D + 28, 12 * D + 16
        # one step = one epoch = `launches_per_step` launches of the update kernel (1; a sharded run whose epoch is cut in segments: that many);
        # the bytes of the schedule and the updates are per EPOCH, so the per-launch figures below are the epoch's divided by the launches
        launches_per_step = max(launches, 1) / max(args.steps, 1)
        epoch_kernel_s = kernel_ms / max(args.steps, 1) / 1e3
        avg_kernel_s = epoch_kernel_s / launches_per_step
        sched = info["schedule_bytes"] / launches_per_step
        n_launch = n_local / launches_per_step
        ach = sched / avg_kernel_s / 1e9                                  # GB/s, rank 0's kernel
        out = {
            "metric": "GloVe pair-updates/sec at dim=%d" % D,
            "value": total_updates / dt,
            "unit": "pair-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16 rows + f32 accumulators (f32 arithmetic)", "data": "synthetic",
            "config": {"workload": ("synthetic hub-heavy co-occurrence matrix (BASELINE C4 recipe scaled to %d GPU%s): " % (world, "s" if world > 1 else "") if K == 1
                                    else "synthetic hub-heavy co-occurrence matrix (BASELINE C4 recipe, the whole %d-GPU job on ONE GPU): " % K) +
                                   "%d vertices, %d nonzeros/GPU, dim=%d, %s cost, %s Hogwild, per-epoch device shuffle"
                                   % (V, n_local, D, args.method, opt.getName()),
                       "vocab": V, "nnz_per_gpu": n_local, "dim": D, "cost": args.method, "opt": args.opt,
                       "parallelism": "rows sharded x%d, context replicated + delta all-reduce every %d step(s) (rows summed, biases averaged, accumulators summed every %d syncs, %s on the wire, %s; hub rows reconciled %s per epoch in fp32)"
                                      % (world, args.sync_every, args.accum_sync_every, args.wire,
                                         "overlapped with the next epoch, %d wavefront slots reserved" % args.reserve_waves if args.exchange == "overlap" else "synchronous",
                                         ("never" if args.hub_segments < 0 else "%d times, %s" % (hub_plan["exchanges"], "beside the running epoch kernel (%d rows live)" % hub_plan["live_rows"] if hub_plan["live"] else "between segments of the epoch")))
                                      if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                         "traffic": traffic,
                         # the counters (FETCH_SIZE x2 + WRITE_SIZE) count L2 <-> fabric requests, Infinity-Cache hits included, and were taken in
                         # the committed profile run (another box, same kernel instance / workload / layout): bytes per launch carry over, the
                         # time divided into them below is THIS run's
                         "traffic_source": "profiles/traffic.json (committed rocprofv3 --pmc passes of this kernel instance, workload and layout)" if traffic else None,
                         "frac_traffic": (traffic / avg_kernel_s / 1e9 / 8000.0) if traffic else None,
                         "kernel": kernel, "kernel_ms": avg_kernel_s * 1e3, "launches_per_step": launches_per_step,
                         "schedule_bytes_per_launch": sched, "schedule_bytes_per_update": sched / max(n_launch, 1),
                         "runs_per_launch": info["runs"] / launches_per_step,
                         "frac_of_measured_copy_ceiling": ach / 6290.0,           # float4 copy, MI355X_MICROARCH.md
                         "box_copy_GBps": box_copy,                # ge_copy_bandwidth on this box, right after the timed region
                         "kernel_updates_per_s": n_launch / avg_kernel_s,
                         # what SURVEY.md 8(d) counts: every update moves both row pairs through HBM (this kernel keeps one in registers)
                         "naive_schedule": {"bytes_per_update": {"read": read_b, "write": write_b},
                                            "bytes_per_launch": n_launch * (read_b + write_b),
                                            "equivalent_GBps": n_launch * (read_b + write_b) / avg_kernel_s / 1e9,
                                            "updates_per_s_over_read_roofline_rate": (n_launch / avg_kernel_s) / (8e12 / read_b)}},
            "mean_cost_first_last": [costs[0] / n_global, costs[-1] / n_global],
            # N > 1: every rank's job costs summed over all nonzeros of the job, per step (warm-up steps first)
            "mean_cost_per_step": [c / n_global for c in costs],
            "trainer": opt.info(),
            # where the driver put the two record tables (which of its two modes a process runs in is fixed when they are placed)
            "table_ptrs": {k: "0x%x" % opt.device_ptr(k)[0] for k in (("focus", "context") if args.dtype == "f32" else ("gsq_focus", "gsq_context"))},
            "gen_seconds": t_gen, "create_seconds": t_create,
        }
        if world > 1:
            out["exchange"] = {"form": args.exchange, "hub_rows": dict(hub_plan, still_live=sync.hub_plan(max(args.hub_segments, 0))["live"]), "kernel_ms_max_over_ranks": kernel_ms_max,
                               "ms_per_step_minus_kernel_ms": dt / args.steps * 1e3 - kernel_ms_max}
            if other:
                out["exchange"]["other_form"] = {"form": other["exchange"], "value": total_updates / other["dt"], "ms_per_step": other["dt"] / args.steps * 1e3,
                                                 "kernel_ms_max_over_ranks": other["kernel_ms"],
                                                 "mean_cost_per_step": [c / n_global for c in other["costs"]],
                                                 "note": "the %d steps right behind the quoted region, same handles" % args.steps}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, V, D, I, J, X, xmax)
        print(json.dumps(out), flush=True)
    if sync is not None:
        sync.close()                  # the RCCL communicator goes before the handle it serves
    opt.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
