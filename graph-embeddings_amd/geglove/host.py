"""Host-side mirror of the reference's interfaces for the hot path (Python flavour, used by
tests and bench.py; the C++ flavour with the `-c config.yml` CLI lives in ../host/).

Every method that computes calls libgeglove.so through ctypes (capi.py).
"""
import ctypes as C
import math
import os

import numpy as np

from . import capi


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class InvalidConfigurationException(Exception):
    """Configuration.InvalidConfigurationException (J/util/config/Configuration.java:496-500)."""

    def __init__(self, message):
        super().__init__("Invalid configuration: " + message)


class GloveCost:
    """J/opt/GloveCost.java:5-21"""
    kind = capi.GE_COST_GLOVE


class PGloveCost:
    """J/opt/PGloveCost.java:5-21"""
    kind = capi.GE_COST_PGLOVE


class Configuration:
    """Tolerant stand-in for the SnakeYAML bean (J/util/config/Configuration.java).

    Accepts the bean's keys AND the legacy keys present in the shipped YAMLs
    (`bca.reverse`, `bca.predicates`, `similarity[].predicate`), which the Java bean of this
    revision would reject.  New, optional keys live under `device:`:
      mode: hogwild|deterministic   shuffle: java|device|none   seed: <long>   id: <ordinal>
      hot: auto|none|all   workers: <int>   dtype: f32|bf16
      hot_theta, stale_budget, flush_every, blocks_per_cu (ge_glove_cfg; 0 / absent = library default)
      layout: [fixed_cuts, plain_long_rows, separate_tables, packed_records, first_placement]   bca_table_slots, bca_pool_entries (ge_bca_cfg sizing)
    """

    def __init__(self, d=None):
        d = dict(d or {})
        self.graph = d.get("graph")
        self.method = d.get("method")
        self.dim = int(d.get("dim", 0) or 0)
        self.threads = int(d.get("threads", 0) or 0)
        self.weights = d.get("weights")
        self.similarity = d.get("similarity")
        self.bca = dict(d.get("bca") or {})
        self.opt = dict(d.get("opt") or {})
        self.pca = d.get("pca")
        self.output = d.get("output")
        self.device = dict(d.get("device") or {})

    @staticmethod
    def load(path):
        """ConfigReader.load (J/util/read/ConfigReader.java:16-20)."""
        import yaml
        with open(path) as f:
            return Configuration(yaml.safe_load(f))

    @staticmethod
    def check(config):
        """Configuration.check (J/util/config/Configuration.java:478-494), same messages."""
        bca = config.bca
        has_bca = bool(bca) and float(bca.get("alpha", 0) or 0) > 0 and float(bca.get("epsilon", 0) or 0) > 0
        out = config.output
        has_out = out is not None and any(k in out and out[k] is not None for k in ("predicate", "blank", "uri", "literal"))
        if not config.dim > 0:
            raise InvalidConfigurationException("No dimension specified")
        if not config.graph:
            raise InvalidConfigurationException("No input graph specified")
        if not config.method:
            raise InvalidConfigurationException("Invalid method, choose one of: glove, pglove")
        if not has_bca:
            raise InvalidConfigurationException("Invalid BCA parameters, alpha and epsilon are mandatory")
        if not has_out:
            raise InvalidConfigurationException("Invalid output parameters, specify at least one group")

    # bean getters used on the hot path
    def getDim(self): return self.dim
    def getThreads(self): return self.threads if self.threads != 0 else max((os.cpu_count() or 2) - 1, 1)   # :71-73
    def getMethod(self): return self.method
    def getAlpha(self): return float(self.bca["alpha"])
    def getEpsilon(self): return float(self.bca["epsilon"])
    def isDirected(self): return bool(self.bca.get("directed", False))
    def getNormalize(self): return (self.bca.get("normalize") or "none")
    def getTolerance(self): return float(self.opt.get("tolerance", 0.0))
    def getMaxiter(self): return int(self.opt.get("maxiter", 0))
    def getOptMethod(self): return self.opt.get("method", "adagrad")

    def costFunction(self):
        """Main.createOptimizer's first switch (J/Main.java:109-119)."""
        m = (self.method or "").upper()
        if m == "GLOVE":
            return GloveCost()
        if m == "PGLOVE":
            return PGloveCost()
        raise ValueError("Invalid cost function")


class CooMatrix:
    """CoOccurrenceMatrix (J/util/CoOccurrenceMatrix.java:6-17) over flat arrays.

    cIdx_* index the matrix in PRE-shuffle order; the epoch permutation lives in the native
    trainer (it is part of the RNG stream there, as in the reference).
    """

    def __init__(self, vocab_size, I, J, X, max_value, keys=None, types=None):
        self._V = int(vocab_size)
        self.I = np.ascontiguousarray(I, np.int32)
        self.J = np.ascontiguousarray(J, np.int32)
        self.X = np.ascontiguousarray(X, np.float32)
        self._max = float(max_value)
        self._keys, self._types = keys, types

    def vocabSize(self): return self._V
    def max(self): return self._max
    def getKey(self, index): return self._keys[index] if self._keys is not None else str(index)
    def getType(self, index): return int(self._types[index]) if self._types is not None else 0
    def cIdx_I(self, i): return int(self.I[i])
    def cIdx_J(self, j): return int(self.J[j])
    def cIdx_C(self, i): return float(self.X[i])
    def coOccurrenceCount(self): return int(self.I.shape[0])
    def shuffle(self): pass  # owned by the native trainer


def _csr_struct(V, csr, keep):
    ptr, idx, w = csr
    ptr = np.ascontiguousarray(ptr, np.int64); idx = np.ascontiguousarray(idx, np.int32)
    w = np.ascontiguousarray(w, np.float32)
    if ptr.shape[0] != V + 1:
        raise ValueError("CSR ptr must have V+1 entries")
    keep.extend([ptr, idx, w])
    return capi.Csr(V, _p(ptr, C.c_int64), _p(idx, C.c_int32), _p(w, C.c_float))


class _CooOwner:
    """Keeps a ge_coo alive for the numpy views of its arrays: every view's buffer object refers to this owner."""

    def __init__(self, handle):
        self._h = handle

    def view(self, ptr, ctype, count, dtype):
        if count <= 0:
            return np.empty(0, dtype)
        buf = (ctype * count).from_address(C.addressof(ptr.contents))
        buf._owner = self
        a = np.frombuffer(buf, dtype=dtype, count=count)
        a.flags.writeable = False
        return a

    def __del__(self):
        try:
            if self._h and self._h.value:
                capi.lib().ge_coo_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class BookmarkColoring(CooMatrix):
    """`new BookmarkColoring(graph, config)` (J/bca/BookmarkColoring.java:32-120) on the device.

    graph: dict(V=int, out=(ptr, idx, w), inn=(ptr, idx, w), keys=[...]?, types=[...]?) --
    the weighted CSR/CSC that getIn/OutNeighborhoods + In/OutEdgeNeighborhoodAlgorithm yield.
    """

    _NORM = {"none": capi.GE_NORM_NONE, "unity": capi.GE_NORM_UNITY, "counts": capi.GE_NORM_COUNTS}

    def __init__(self, graph, config, device=0, row_range=None):
        V = int(graph["V"])
        keep = []
        out_s = _csr_struct(V, graph["out"], keep)
        in_s = _csr_struct(V, graph["inn"], keep)
        norm = self._NORM[str(config.getNormalize()).lower()]
        rb, re = row_range if row_range else (0, 0)
        dev = config.device or {}
        cfg = capi.BcaCfg(config.getAlpha(), config.getEpsilon(), int(config.isDirected()), norm, device, rb, re,
                          int(dev.get("bca_table_slots", 0)), int(dev.get("bca_pool_entries", 0)))
        h = C.c_void_p()
        capi.check(capi.lib().ge_bca_build(C.byref(out_s), C.byref(in_s), C.byref(cfg), C.byref(h)))
        owner = _CooOwner(h)                 # the arrays below are VIEWS of the library's result (780 MB at 65 M entries: no copies); it is
        nnz = C.c_int64(); mx = C.c_double()  # destroyed when the last of them goes
        pI = C.POINTER(C.c_int32)(); pJ = C.POINTER(C.c_int32)(); pX = C.POINTER(C.c_float)()
        pR = C.POINTER(C.c_int64)()
        capi.check(capi.lib().ge_coo_get(h, C.byref(nnz), C.byref(pI), C.byref(pJ), C.byref(pX), C.byref(pR), C.byref(mx)))
        n = nnz.value
        I = owner.view(pI, C.c_int32, n, np.int32)
        J = owner.view(pJ, C.c_int32, n, np.int32)
        X = owner.view(pX, C.c_float, n, np.float32)
        self.row_ptr = owner.view(pR, C.c_int64, V + 1, np.int64).copy()
        super().__init__(V, I, J, X, mx.value, graph.get("keys"), graph.get("types"))


class Optimum:
    """J/opt/Optimum.java:9-41"""

    def __init__(self):
        self.finalCost = 0.0
        self.result = None
        self.costHistory = []

    def addIntermediaryResult(self, r): self.costHistory.append(r)
    def getResult(self): return self.result
    def setResult(self, r): self.result = r
    def getFinalCost(self): return self.finalCost
    def setFinalCost(self, c): self.finalCost = c


_MODES = {"hogwild": capi.GE_MODE_HOGWILD, "deterministic": capi.GE_MODE_DETERMINISTIC}
_HOT = {"auto": capi.GE_HOT_AUTO, "none": capi.GE_HOT_NONE, "all": capi.GE_HOT_ALL}
_SHUFFLES = {"java": capi.GE_SHUFFLE_JAVA, "device": capi.GE_SHUFFLE_DEVICE, "none": capi.GE_SHUFFLE_NONE}
_LAYOUT = {"default": 0, "fixed_cuts": capi.GE_LAYOUT_FIXED_CUTS, "plain_long_rows": capi.GE_LAYOUT_PLAIN_LONG_ROWS,
           "separate_tables": capi.GE_LAYOUT_SEPARATE_TABLES, "packed_records": capi.GE_LAYOUT_PACKED_RECORDS,
           "first_placement": capi.GE_LAYOUT_FIRST_PLACEMENT}


class Adagrad:
    OPT = capi.GE_OPT_ADAGRAD
    NAME = "Adagrad"
    """`new Adagrad(coMatrix, config, costFunction)` + IOptimizer (J/opt/grad/Adagrad.java,
    J/opt/Optimizer.java, J/opt/IOptimizer.java:6-11) over ge_glove_*.

    Extra keyword arguments override config.device.* : mode, shuffle, seed, device, stream,
    row_range (multi-GPU sharding), learning_rate.
    """

    def __init__(self, coMatrix, config, costFunction, **kw):
        dev = dict(config.device)
        dev.update({k: v for k, v in kw.items() if v is not None})
        self.coMatrix, self.config, self.costFunction = coMatrix, config, costFunction
        self.dimension = config.getDim()
        self.vocabSize = coMatrix.vocabSize()
        self.coCount = coMatrix.coOccurrenceCount()
        self.numThreads = config.getThreads()
        self.maxIterations = config.getMaxiter()
        self.tolerance = config.getTolerance()
        cfg = capi.GloveCfg()
        capi.lib().ge_glove_cfg_default(C.byref(cfg))
        cfg.vocab_size, cfg.dim, cfg.nnz = self.vocabSize, self.dimension, self.coCount
        cfg.cost = costFunction.kind
        cfg.opt = self.OPT
        cfg.learning_rate = float(dev.get("learning_rate", 0.05))
        cfg.xmax = coMatrix.max()
        cfg.seed = int(dev.get("seed", 42))
        cfg.threads = self.numThreads
        cfg.mode = _MODES[str(dev.get("mode", "hogwild")).lower()]
        cfg.shuffle = _SHUFFLES[str(dev.get("shuffle", "device")).lower()]
        cfg.device = int(dev.get("id", dev.get("device", 0)))
        cfg.stream = dev.get("stream", None)
        rb, re = dev.get("row_range", (0, 0))
        cfg.row_begin, cfg.row_end = rb, re
        cfg.hot_columns = _HOT[str(dev.get("hot", "auto")).lower()]
        cfg.workers = int(dev.get("workers", 0))
        cfg.emb_dtype = {"f32": capi.GE_DTYPE_F32, "bf16": capi.GE_DTYPE_BF16}[str(dev.get("dtype", "f32")).lower()]
        cfg.hot_theta = float(dev.get("hot_theta", 0)); cfg.stale_budget = float(dev.get("stale_budget", 0))
        cfg.flush_every = int(dev.get("flush_every", 0)); cfg.blocks_per_cu = int(dev.get("blocks_per_cu", 0))
        lay = dev.get("layout") or []
        for name in ([lay] if isinstance(lay, str) else lay):
            cfg.layout_flags |= _LAYOUT[str(name).lower()]
        self._rows = (re - rb) if (rb, re) != (0, 0) else self.vocabSize
        self._cfg = cfg
        self._h = C.c_void_p()
        I, J, X = coMatrix.I, coMatrix.J, coMatrix.X
        capi.check(capi.lib().ge_glove_create(C.byref(cfg), _p(I, C.c_int32), _p(J, C.c_int32), _p(X, C.c_float),
                                              C.byref(self._h)))

    # -- IOptimizer -----------------------------------------------------------------------
    def getName(self):
        return self.NAME

    def epoch(self, iteration=0):
        """One loop body of Optimizer.optimize (shuffle + all jobs); returns the summed job cost."""
        cost = C.c_double()
        capi.check(capi.lib().ge_glove_epoch(self._h, iteration, C.byref(cost)))
        return cost.value

    def createJob(self, id, iteration):
        """Natively one call runs every job of the epoch; job 0 carries it, the others are empty."""
        if id == 0:
            return lambda: self.epoch(iteration)
        return lambda: 0.0

    def optimize(self):
        """Optimizer.optimize (J/opt/Optimizer.java:66-120): tolerance / history logic stays on the host."""
        opt = Optimum()
        finalCost = 0.0
        prevCost = 0.0
        for iteration in range(self.maxIterations):
            localCost = self.epoch(iteration)
            localCost = localCost / self.coCount if self.coCount else float("nan")
            opt.addIntermediaryResult(localCost)
            iterDiff = abs(prevCost - localCost)
            prevCost = localCost
            if iterDiff <= self.tolerance:
                finalCost = localCost
                break
        opt.setResult(self.extractResult())
        opt.setFinalCost(finalCost)
        return opt

    def extractResult(self):
        out = np.empty(self.vocabSize * self.dimension, np.float64)
        capi.check(capi.lib().ge_glove_extract_f64(self._h, _p(out, C.c_double)))
        return out

    def extractResultF32(self):
        out = np.empty(self.vocabSize * self.dimension, np.float32)
        capi.check(capi.lib().ge_glove_extract_f32(self._h, _p(out, C.c_float)))
        return out

    # -- state access (tests / multi-GPU sync) ---------------------------------------------------
    def _count(self, which):
        rows = self._rows if which in (capi.GE_STATE_FOCUS, capi.GE_STATE_FBIAS, capi.GE_STATE_GSQ_FOCUS,
                                       capi.GE_STATE_GSQ_FBIAS, capi.GE_STATE_M2_FOCUS, capi.GE_STATE_M2_FBIAS) else self.vocabSize
        per_row = self.dimension if which in (capi.GE_STATE_FOCUS, capi.GE_STATE_CONTEXT, capi.GE_STATE_GSQ_FOCUS,
                                              capi.GE_STATE_GSQ_CONTEXT, capi.GE_STATE_M2_FOCUS, capi.GE_STATE_M2_CONTEXT) else 1
        return rows * per_row

    def get_state(self, which):
        if isinstance(which, str):
            which = capi.ALL_STATE_NAMES.index(which)
        n = self._count(which)
        out = np.empty(n, np.float32)
        capi.check(capi.lib().ge_glove_get_state(self._h, which, _p(out, C.c_float), n))
        return out

    def set_state(self, which, arr):
        if isinstance(which, str):
            which = capi.ALL_STATE_NAMES.index(which)
        arr = np.ascontiguousarray(arr, np.float32).reshape(-1)
        capi.check(capi.lib().ge_glove_set_state(self._h, which, _p(arr, C.c_float), arr.shape[0]))

    def state(self):
        names = capi.STATE_NAMES if self.OPT == capi.GE_OPT_ADAGRAD else capi.ALL_STATE_NAMES
        return {n: self.get_state(i) for i, n in enumerate(names)}

    def device_ptr(self, which):
        if isinstance(which, str):
            which = capi.ALL_STATE_NAMES.index(which)
        p = C.c_void_p(); n = C.c_int64()
        capi.check(capi.lib().ge_glove_device_ptr(self._h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def perm(self):
        out = np.empty(self.coCount, np.int32)
        capi.check(capi.lib().ge_glove_get_perm(self._h, _p(out, C.c_int32), self.coCount))
        return out

    def epoch_order(self, iteration):
        """Order in which a single worker walks the nonzeros in epoch `iteration` (ge_glove_epoch_order)."""
        out = np.empty(self.coCount, np.int32)
        capi.check(capi.lib().ge_glove_epoch_order(self._h, iteration, _p(out, C.c_int32), self.coCount))
        return out

    def rng_state(self):
        s = C.c_uint64()
        capi.check(capi.lib().ge_glove_rng_state(self._h, C.byref(s)))
        return s.value

    def last_kernel_ms(self):
        ms = C.c_float(); n = C.c_int32()
        capi.check(capi.lib().ge_glove_last_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def info(self):
        inf = capi.GloveInfo()
        capi.check(capi.lib().ge_glove_get_info(self._h, C.byref(inf)))
        return {k: getattr(inf, k) for k, _ in capi.GloveInfo._fields_}

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            capi.lib().ge_glove_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Adam(Adagrad):
    """`new Adam(coMatrix, config, costFunction)` (J/opt/grad/Adam.java)"""
    OPT = capi.GE_OPT_ADAM
    NAME = "Adam"


class AMSGrad(Adagrad):
    """`new AMSGrad(coMatrix, config, costFunction)` (J/opt/grad/AMSGrad.java)"""
    OPT = capi.GE_OPT_AMSGRAD
    NAME = "AMSGrad"


def createOptimizer(config, coMatrix, **kw):
    """Main.createOptimizer (J/Main.java:107-131)."""
    cf = config.costFunction()
    m = str(config.getOptMethod()).upper()
    if m == "ADAGRAD":
        return Adagrad(coMatrix, config, cf, **kw)
    if m == "ADAM":
        return Adam(coMatrix, config, cf, **kw)
    if m == "AMSGRAD":
        return AMSGrad(coMatrix, config, cf, **kw)
    raise ValueError("Invalid optimization method")


# ------------------------------------------------------------------------------------------------
# literal-similarity edges
# ------------------------------------------------------------------------------------------------
SIMILARITY_METHODS = ("ngram_cosine", "ngram_jaccard", "token_cosine", "token_jaccard", "jarowinkler", "levenshtein",
                      "numeric", "date_days", "date_months", "date_years")      # Configuration.SimilarityMethod (:27-29)
SIMILARITY_TIMES = ("backwards", "forwards", "bidirectional")                    # SimilarityGroup.Time (:184-186)


class SimilarityGroup:
    """Configuration.SimilarityGroup (J/util/config/Configuration.java:182-318): one `similarity:` entry of the YAML."""

    def __init__(self, d):
        self.sourcePredicate = d.get("sourcePredicate", d.get("predicate"))
        self.targetPredicate = d.get("targetPredicate", d.get("predicate"))
        self.method = d["method"]
        self.threshold = float(d.get("threshold", 0.0))
        self.ngram = int(d.get("ngram", 0))
        self.distance = float(d.get("distance", 0.0))
        self.smooth = float(d.get("smooth", 0.0))
        self.pattern = d.get("pattern")
        self.time = d.get("time")

    def getMethodEnum(self):
        m = self.method.lower()
        if m not in SIMILARITY_METHODS:
            raise ValueError("No enum constant SimilarityMethod." + self.method.upper())      # valueOf throws
        return SIMILARITY_METHODS.index(m)

    def getNgram(self): return 3 if self.ngram == 0 else self.ngram
    def getSmooth(self): return 1.0 if self.smooth == 0 else self.smooth
    def getPattern(self): return "iso" if self.pattern is None else self.pattern
    def getTime(self): return "bidirectional" if self.time is None else self.time
    def getTimeEnum(self): return SIMILARITY_TIMES.index(self.getTime().lower())


class CompareGroup:
    """CompareGroup + the CompareJob loop (J/compare/CompareGroup.java, CompareJob.java, Rdf2GrphConverter.java:127-186)
    on the device: compare(labels) returns the (source vertex, target vertex, float similarity) triples for which the
    reference adds its two directed edges, in job order."""

    def __init__(self, group, device=0):
        self.group = group
        self.upperTriangle = group.sourcePredicate == group.targetPredicate       # Rdf2GrphConverter.java:51
        self.source, self.target = [], []
        self.device = device

    def addToSource(self, vertex): self.source.append(int(vertex))
    def addToTarget(self, vertex): self.target.append(int(vertex))

    def compare(self, vertex_labels, job_range=None):
        """vertex_labels: vertex id -> label (g.getVertexLabelProperty()).  job_range = (begin, end): only those
        CompareJobs (source positions) run -- the shard of one GPU; concatenating the shards gives the whole result."""
        g = self.group
        verts = sorted(set(self.source) | set(self.target))
        pos = {v: k for k, v in enumerate(verts)}
        parts = [np.frombuffer(vertex_labels[v].encode("utf-16-le"), dtype=np.uint16) for v in verts]
        off = np.zeros(len(parts) + 1, np.int64)
        if parts:
            off[1:] = np.cumsum([len(x) for x in parts])
        units = np.ascontiguousarray(np.concatenate(parts) if parts and off[-1] else np.zeros(1, np.uint16), dtype=np.uint16)
        table = capi.Strings(len(parts), _p(off, C.c_int64), _p(units, C.c_uint16))
        sv = np.ascontiguousarray(self.source, dtype=np.int32); tv = np.ascontiguousarray(self.target, dtype=np.int32)
        sp = np.ascontiguousarray([pos[v] for v in self.source], dtype=np.int32)
        tp = np.ascontiguousarray([pos[v] for v in self.target], dtype=np.int32)
        cfg = capi.SimCfg()
        L = capi.lib()
        L.ge_sim_cfg_default(C.byref(cfg))
        cfg.method = g.getMethodEnum(); cfg.threshold = g.threshold; cfg.ngram = g.getNgram(); cfg.smooth = g.getSmooth()
        cfg.distance = g.distance; cfg.time = g.getTimeEnum()
        cfg.pattern = None if g.getPattern() == "iso" else g.getPattern().encode()
        cfg.upper_triangle = int(self.upperTriangle); cfg.device = self.device
        if job_range is not None:
            cfg.job_begin, cfg.job_end = int(job_range[0]), int(job_range[1])
        h = C.c_void_p()
        capi.check(L.ge_similarity_pairs(C.byref(table), _p(sp, C.c_int32), _p(sv, C.c_int32), len(sp),
                                         _p(tp, C.c_int32), _p(tv, C.c_int32), len(tp), C.byref(cfg), C.byref(h)))
        try:
            n = C.c_int64(); pi = C.POINTER(C.c_int32)(); pj = C.POINTER(C.c_int32)(); ps = C.POINTER(C.c_float)()
            capi.check(L.ge_sim_pairs_get(h, C.byref(n), C.byref(pi), C.byref(pj), C.byref(ps)))
            k = n.value
            i = np.ctypeslib.as_array(pi, (k,)).copy() if k else np.zeros(0, np.int32)
            j = np.ctypeslib.as_array(pj, (k,)).copy() if k else np.zeros(0, np.int32)
            sim = np.ctypeslib.as_array(ps, (k,)).copy() if k else np.zeros(0, np.float32)
        finally:
            L.ge_sim_pairs_destroy(h)
        self.pairs = (i, j, sim)                                       # positions in source / target, as CompareResult holds them
        return sv[i], tv[j], sim
