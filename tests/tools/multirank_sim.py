"""CPU simulation of the multi-GPU merge rules: W simulated ranks, each running the oracle's
Adagrad.createJob restatement on its row shard against its own replica of the context side, merged once per epoch.

  python tests/tools/multirank_sim.py [--ranks 8] [--epochs 14] [--delay 0|1] ...

--delay 1 applies the other ranks' summed deltas one epoch late (the overlapped exchange of ContextSync.begin/finish:
the all-reduce of epoch k's deltas runs under epoch k+1).  Prints mean cost per epoch divided by the single-process
oracle's.  Uses oracle/ as the stand-in for the device pass: a design tool, not part of the product.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "graph-embeddings_amd"))
import oracle as O                      # noqa: E402
from geglove import parallel, synth     # noqa: E402

SUMS, MEANS = ("context", "gsq_context", "gsq_cbias"), ("cbias",)


def bf16(a):
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def model_cost(I, J, X, xmax, focus, context, fb, cb, block=200_000):
    """GloVe cost of a fixed model over all nonzeros (GloveCost.java:9-20 without the update), mean per nonzero."""
    tot = 0.0
    for a in range(0, len(I), block):
        i, j, x = I[a:a + block], J[a:a + block], X[a:a + block].astype(np.float64)
        inner = np.einsum("nd,nd->n", focus[i].astype(np.float64), context[j].astype(np.float64)) + fb[i] + cb[j] - np.log(x)
        w = np.minimum(1.0, (x / xmax) ** 0.75)
        tot += float(np.sum(0.5 * w * inner * inner))
    return tot / len(I)


def run(V, N, D, W, epochs, delay, wire, seed=13, accum_every=1, hub_segments=0, hub_workers=180, merge="adagrad"):
    I, J, X, xmax = synth.synthetic_coo(V, N, seed=seed)
    N = len(I)                                             # duplicates are merged by the generator
    single = O.Glove(V, D, I, J, X, xmax, O.COST_GLOVE, seed=42, threads=1)
    init = {k: v.copy() for k, v in single.state().items()}
    rng = np.random.default_rng(5)
    ref = []
    st1 = {k: v.copy() for k, v in init.items()}
    for _ in range(epochs):
        p = rng.permutation(N)
        O.adagrad_job(D, I[p], J[p], X[p], xmax, O.COST_GLOVE, st1)
        ref.append(model_cost(I, J, X, xmax, st1["focus"].reshape(V, D), st1["context"].reshape(V, D), st1["fbias"], st1["cbias"]))

    shards = [parallel.shard_nonzeros(I, J, X, parallel.shard_rows(V, W, r)) for r in range(W)]
    rows = [parallel.shard_rows(V, W, r) for r in range(W)]
    st = [{k: v.copy() for k, v in init.items()} for _ in range(W)]
    base = [{k: st[r][k].copy() for k in SUMS + MEANS} for r in range(W)]
    pending = None
    out = []
    rngs = [np.random.default_rng(100 + r) for r in range(W)]
    narrow = bf16 if wire == "bf16" else (lambda a: a)
    for e in range(epochs):
        tot = 0.0
        if hub_segments:
            # ge_sync_epoch: every rank's pass in S segments; behind each, the HUB rows (union of the ranks' hub columns: count on a
            # rank >= 0.25 N_rank / workers) are reconciled exactly -- rows and accumulators summed, cBias averaged over its movers
            if e == 0:
                hub = np.zeros(V, bool)
                for r in range(W):
                    c = np.bincount(shards[r][1], minlength=V)
                    hub |= c >= max(1, int(0.25 * len(shards[r][0]) / hub_workers))
                hubs = np.nonzero(hub)[0]
                print("hub rows: %d of %d columns, %.1f %% of the nonzeros" % (len(hubs), V, 100.0 * hub[J].mean()), flush=True)
            perms = [rngs[r].permutation(len(shards[r][0])) for r in range(W)]
            for sgm in range(hub_segments):
                for r in range(W):
                    si, sj, sx = shards[r]
                    p = perms[r][len(si) * sgm // hub_segments:len(si) * (sgm + 1) // hub_segments]
                    tot += float(O.adagrad_job(D, si[p], sj[p], sx[p], xmax, O.COST_GLOVE, st[r]))
                g0 = base[0]["gsq_context"].reshape(V, -1)[hubs].copy()
                e_sum = np.maximum(sum(st[r]["gsq_context"].reshape(V, -1)[hubs] - g0 for r in range(W)), 0.0)
                for k in SUMS + MEANS:
                    cur = [st[r][k].reshape(V, -1)[hubs] for r in range(W)]
                    b0 = base[0][k].reshape(V, -1)[hubs]
                    d = [c - b0 for c in cur]
                    tsum = sum(d)
                    if k in MEANS:
                        tsum = tsum / np.maximum(sum((x != 0).astype(np.float32) for x in d), 1.0)
                    if k == "context" and merge == "adagrad":       # csrc/sync.hip merge_scale
                        tsum = np.sqrt((g0 + e_sum / W) / (g0 + e_sum)).astype(np.float32) * tsum
                    new = b0 + tsum
                    for r in range(W):
                        st[r][k].reshape(V, -1)[hubs] = new
                        base[r][k].reshape(V, -1)[hubs] = new
        else:
          for r in range(W):
            si, sj, sx = shards[r]
            p = rngs[r].permutation(len(si))
            tot += float(O.adagrad_job(D, si[p], sj[p], sx[p], xmax, O.COST_GLOVE, st[r]))
        # snapshot this epoch's deltas
        own = [{k: (narrow(st[r][k] - base[r][k]) if k in ("context", "gsq_context") else st[r][k] - base[r][k])
                for k in SUMS + MEANS} for r in range(W)]
        for r in range(W):
            for k in SUMS + MEANS:
                if k.startswith("gsq") and (e + 1) % accum_every != 0:
                    continue
                base[r][k] = st[r][k].copy()
        merged = {}
        acc_due = (e + 1) % accum_every == 0
        for k in SUMS:
            merged[k] = sum(own[r][k] for r in range(W))
            if k.startswith("gsq") and not acc_due:            # accumulators not exchanged this step: nothing lands, nothing is taken
                merged[k] = None
        for k in MEANS:
            cnt = sum((own[r][k] != 0).astype(np.float32) for r in range(W))
            merged[k] = sum(own[r][k] for r in range(W)) / np.maximum(cnt, 1.0)
        ready = (merged, own)
        if delay:
            ready, pending = pending, ready
        if ready is not None:
            m, o = ready
            for r in range(W):
                for k in SUMS + MEANS:
                    if m[k] is None:
                        continue
                    R = m[k] - o[r][k]
                    st[r][k] += R
                    base[r][k] += R
        # the merged model: every focus row from its owner, the context side as it will be once everything in
        # flight has landed (rank 0's replica plus what it has not received yet)
        focus = np.concatenate([st[r]["focus"].reshape(V, D)[rows[r][0]:rows[r][1]] for r in range(W)])
        fb = np.concatenate([st[r]["fbias"][rows[r][0]:rows[r][1]] for r in range(W)])
        ctx, cb = st[0]["context"].copy(), st[0]["cbias"].copy()
        if pending is not None:
            ctx += pending[0]["context"] - pending[1][0]["context"]
            cb += pending[0]["cbias"] - pending[1][0]["cbias"]
        out.append(model_cost(I, J, X, xmax, focus, ctx.reshape(V, D), fb, cb))
    return np.array(out) / np.array(ref), ref


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=4000)
    ap.add_argument("--nnz", type=int, default=160000)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=14)
    ap.add_argument("--wire", default="bf16")
    ap.add_argument("--accum-every", type=int, default=1)
    ap.add_argument("--delays", default="0,1")
    ap.add_argument("--hub-segments", type=int, default=0, help="S > 0: ge_sync_epoch -- the hub rows are reconciled S times per epoch")
    ap.add_argument("--hub-workers", type=int, default=180, help="workers of the hub threshold 0.25 N_rank / workers")
    ap.add_argument("--merge", default="adagrad", choices=["adagrad", "sum"], help="the hub rows' summed deltas: scaled by sqrt((G0 + E / W) / (G0 + E)) of the accumulators (the library) or summed as they are")
    a = ap.parse_args()
    for delay in [int(x) for x in a.delays.split(",")]:
        ratio, ref = run(a.vocab, a.nnz, a.dim, a.ranks, a.epochs, delay, a.wire, accum_every=a.accum_every, hub_segments=a.hub_segments, hub_workers=a.hub_workers, merge=a.merge)
        print("delay %d:" % delay, " ".join("%.3f" % x for x in ratio), flush=True)
    print("single-process cost:", " ".join("%.4f" % x for x in ref))
