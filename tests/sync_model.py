"""A model of the multi-GPU context exchange in plain torch ops on HOST tensors (test infrastructure).

The product's exchange lives behind the C ABI (ge_sync_*, graph-embeddings_amd/csrc/sync.hip: device kernels + RCCL).  This
file restates its merge rule -- deltas of the context rows and the AdaGrad accumulators add, cBias takes the mean over the ranks
that moved the element, accumulators only every lazy_every-th exchange, take / land with a base that carries what is in flight --
so that (a) world_size-2 gloo tests can run the N > 1 logic on a machine without a GPU and (b) the GPU tests have something
independent to hold ge_sync against.  It never runs in the product."""
import torch
import torch.distributed as dist


class SyncModel:
    def __init__(self, sums, means, lazy_sums=(), lazy_every=4, wire="bf16", group=None, merge="adagrad"):
        self.group = group
        self.merge = merge                   # how the hub rows' summed deltas are merged: "adagrad" (csrc/sync.hip merge_scale) or "sum"
        self.world = dist.get_world_size(group)
        self.wire = wire
        self.lazy_every = max(1, int(lazy_every))
        self.calls = 0
        self.ent = ([dict(t=t, o=t.clone(), mean=False, lazy=False, work=None) for t in sums] +
                    [dict(t=t, o=t.clone(), mean=False, lazy=True, work=None) for t in lazy_sums] +
                    [dict(t=t, o=t.clone(), mean=True, lazy=False, work=None) for t in means])
        # a row table's accumulator table: the lazy sum of the same size (context <-> gradSqContext)
        for e in self.ent:
            e["acc"] = None
            if not e["mean"] and not e["lazy"]:
                for a in self.ent:
                    if a["lazy"] and a["t"].numel() == e["t"].numel():
                        e["acc"] = a
        for e in self.ent:
            narrow = wire == "bf16" and not e["mean"]
            e["w"] = torch.empty(e["t"].shape, dtype=torch.bfloat16 if narrow else e["t"].dtype)
            e["own"] = torch.empty_like(e["w"]); e["cnt"] = None

    def _turn(self, land, take, everything=False):
        """o = the consensus c, own = this rank's delta in flight (csrc/sync.hip, k_sync_turn)."""
        if self.world == 1:
            return
        due = False
        if take:
            self.calls += 1
            due = everything or self.calls % self.lazy_every == 0
        for e in self.ent:
            do_land = land and e["work"] is not None
            do_take = take and (due or not e["lazy"])
            if not (do_land or do_take):
                continue
            t, c, w, own = e["t"], e["o"], e["w"], e["own"]
            resid = t - c
            if do_land:
                for x in e["work"]:
                    x.wait()
                e["work"] = None
                m = w.to(torch.float32)
                if e["mean"]:
                    m = m / e["cnt"].clamp(min=1.0)
                c.add_(m)
                resid = resid - own.to(torch.float32)
                t.copy_(c + resid)
            if do_take:
                w.copy_(resid); own.copy_(w)
                work = []
                if e["mean"]:
                    e["cnt"] = own.ne(0).to(torch.float32)
                    work.append(dist.all_reduce(e["cnt"], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                work.append(dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                e["work"] = work

    def _merge_scale(self, g0, e_sum):
        """csrc/sync.hip merge_scale: sqrt((G0 + E / W) / (G0 + E)) in fp32, operation for operation (1 for the plain sum)."""
        if self.merge != "adagrad":
            return torch.ones_like(g0)
        inv_w = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(float(self.world), dtype=torch.float32)
        e = torch.clamp(e_sum, min=0.0)
        q = (g0 + e * inv_w) / (g0 + e)
        return torch.sqrt(q.double()).float()             # (torch's fp32 sqrt on the CPU is a vector routine that is not correctly rounded; the fp64 one, rounded once more, is)

    def hub_exchange(self, rows, n_rows):
        """csrc/sync.hip k_hub_take / k_hub_land: the rows `rows` (of n_rows) of EVERY table reconciled now, in fp32 and exactly:
        delta = table - c, summed over the ranks (means: divided by the number of ranks that moved the element; a row table: scaled
        by merge_scale of its accumulator table's consensus and summed delta), c += that, table = c."""
        if self.world == 1 or len(rows) == 0:
            return
        idx = torch.as_tensor(rows, dtype=torch.long)
        tot = {}
        for e in self.ent:                                        # every table's summed delta first: the merge of a row table reads
            t, c = e["t"].view(n_rows, -1), e["o"].view(n_rows, -1)   # its accumulator table's
            d = (t[idx] - c[idx]).contiguous()
            if e["mean"]:
                cnt = d.ne(0).to(torch.float32)
                dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(d, op=dist.ReduceOp.SUM, group=self.group)
            if e["mean"]:
                d = d / cnt.clamp(min=1.0)
            tot[id(e)] = d
        scale = {id(e): self._merge_scale(e["acc"]["o"].view(n_rows, -1)[idx], tot[id(e["acc"])]) for e in self.ent if e["acc"] is not None}
        for e in self.ent:
            t, c = e["t"].view(n_rows, -1), e["o"].view(n_rows, -1)
            d = tot[id(e)]
            if id(e) in scale:
                d = scale[id(e)] * d
            new = c[idx] + d
            c[idx] = new
            t[idx] = new

    def hub_exchange_live(self, rows, n_rows, later=None):
        """csrc/sync.hip k_live_take / k_live_land: the live exchange of the rows `rows` of the SUM tables that are not lazy-only scalars
        (the context rows and their accumulator rows: what the epoch kernel moves by atomic adds).  own = table - c at the take; the
        table may move on while the sum is under way (`later(e, idx)` returns what this rank adds meanwhile, or None); the land ADDS
        merged - own to whatever the table holds then and merged to c (merged = the sum; a row table's: scaled by merge_scale)."""
        if self.world == 1 or len(rows) == 0:
            return
        idx = torch.as_tensor(rows, dtype=torch.long)
        live = [e for e in self.ent if not e["mean"] and e["t"].view(n_rows, -1).shape[1] > 1]      # the rows' scalars wait for the exact exchange
        own, total = {}, {}
        for e in live:                                            # take + all-reduce of both tables ...
            t, c = e["t"].view(n_rows, -1), e["o"].view(n_rows, -1)
            own[id(e)] = (t[idx] - c[idx]).contiguous()
            total[id(e)] = own[id(e)].clone()
            dist.all_reduce(total[id(e)], op=dist.ReduceOp.SUM, group=self.group)
        scale = {id(e): self._merge_scale(e["acc"]["o"].view(n_rows, -1)[idx], total[id(e["acc"])]) for e in live if e["acc"] is not None}
        for e in live:                                            # ... then the land (the row table's merge reads the accumulators' consensus BEFORE it moves)
            t, c = e["t"].view(n_rows, -1), e["o"].view(n_rows, -1)
            if later is not None:
                mv = later(e, idx)
                if mv is not None:
                    t[idx] = t[idx] + mv
            merged = scale[id(e)] * total[id(e)] if id(e) in scale else total[id(e)]
            e["_land"] = (merged, own[id(e)])
        for e in live:
            t, c = e["t"].view(n_rows, -1), e["o"].view(n_rows, -1)
            merged, o = e.pop("_land")
            t[idx] = t[idx] + (merged - o)
            c[idx] = c[idx] + merged

    def begin(self, everything=False): self._turn(False, True, everything)
    def finish(self): self._turn(True, False)
    def turn(self, everything=False): self._turn(True, True, everything)
    def sync(self): self.turn(); self.finish()

    def replicate(self, src=0):
        if self.world == 1:
            return
        self.turn(everything=True)
        self.finish()
        for e in self.ent:
            dist.broadcast(e["t"], src=src, group=self.group)
            e["o"].copy_(e["t"])
