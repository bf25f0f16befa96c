"""ctypes wrapper around oracle/libge_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never from the product package.
PARITY UNPINNED (no JDK, no reference fixtures): see ge_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GE_ORACLE_LIB: another build of the same sources (oracle/Makefile's `asan` target: -fsanitize=address,undefined)
_LIB_PATH = os.environ.get("GE_ORACLE_LIB") or os.path.join(_HERE, "libge_oracle.so")

NORM_NONE, NORM_UNITY, NORM_COUNTS = 0, 1, 2
COST_GLOVE, COST_PGLOVE = 0, 1
OPT_ADAGRAD, OPT_ADAM, OPT_AMSGRAD = 0, 1, 2
STATE_NAMES = ("focus", "context", "fbias", "cbias", "gsq_focus", "gsq_context", "gsq_fbias", "gsq_cbias",
               "m2_focus", "m2_context", "m2_fbias", "m2_cbias")


SIM_METHODS = ("ngram_cosine", "ngram_jaccard", "token_cosine", "token_jaccard", "jarowinkler", "levenshtein", "numeric",
               "date_days", "date_months", "date_years")          # Configuration.SimilarityMethod ordinals
SIM_TIMES = ("backwards", "forwards", "bidirectional")


class SimCfg(C.Structure):
    _fields_ = [("method", C.c_int32), ("threshold", C.c_double), ("ngram", C.c_int32), ("smooth", C.c_double),
                ("distance", C.c_double), ("time", C.c_int32), ("pattern", C.c_char_p)]


def sim_cfg(method, threshold=0.0, ngram=3, smooth=1.0, distance=0.0, time="bidirectional", pattern=None):
    return SimCfg(SIM_METHODS.index(method.lower()), float(threshold), int(ngram) or 3, float(smooth) or 1.0, float(distance),
                  SIM_TIMES.index(time.lower()), None if pattern is None else pattern.encode())


def utf16(s):
    """java.lang.String view of a Python str: UTF-16 code units."""
    return np.frombuffer(s.encode("utf-16-le"), dtype=np.uint16).copy() if s else np.zeros(0, np.uint16)


def string_table(strings):
    """(offset int64[n+1], units uint16[...]) of a list of Python strings."""
    parts = [utf16(s) for s in strings]
    off = np.zeros(len(parts) + 1, np.int64)
    off[1:] = np.cumsum([len(p) for p in parts])
    units = np.concatenate(parts) if parts and off[-1] else np.zeros(1, np.uint16)
    return off, np.ascontiguousarray(units, dtype=np.uint16)


def sim_pair(cfg, s1, s2):
    """metric.similarity(s1, s2); returns (value, threw)."""
    a, b = utf16(s1), utf16(s2)
    a_ = np.ascontiguousarray(a if len(a) else np.zeros(1, np.uint16)); b_ = np.ascontiguousarray(b if len(b) else np.zeros(1, np.uint16))
    threw = C.c_int(0)
    v = lib().geo_sim_pair(C.byref(cfg), _p(a_, C.c_uint16), len(a), _p(b_, C.c_uint16), len(b), C.byref(threw))
    return v, bool(threw.value)


def compare_group(cfg, strings, source, target, source_vertex=None, target_vertex=None, upper_triangle=False):
    """The CompareJob loop of one CompareGroup with threads: 1 -> (i, j, float32 similarity) arrays."""
    off, units = string_table(strings)
    src = np.ascontiguousarray(source, dtype=np.int32); tgt = np.ascontiguousarray(target, dtype=np.int32)
    sv = np.ascontiguousarray(src if source_vertex is None else source_vertex, dtype=np.int32)
    tv = np.ascontiguousarray(tgt if target_vertex is None else target_vertex, dtype=np.int32)
    args = (C.byref(cfg), _p(off, C.c_int64), _p(units, C.c_uint16), _p(src, C.c_int32), _p(sv, C.c_int32), len(src),
            _p(tgt, C.c_int32), _p(tv, C.c_int32), len(tgt), int(bool(upper_triangle)))
    cap = 1 << 16
    while True:
        oi = np.zeros(cap, np.int32); oj = np.zeros(cap, np.int32); osim = np.zeros(cap, np.float32)
        n = lib().geo_compare_group(*args, _p(oi, C.c_int32), _p(oj, C.c_int32), _p(osim, C.c_float), cap)
        if n <= cap:
            return oi[:n].copy(), oj[:n].copy(), osim[:n].copy()
        cap = int(n)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("ge_oracle.c", "ge_oracle_sim.c", "ge_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["asan"] if _LIB_PATH.endswith("_asan.so") else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    i32p, i64p, f32p, f64p = (C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                              C.POINTER(C.c_float), C.POINTER(C.c_double))
    L.geo_jrand_init.argtypes = [C.c_void_p, C.c_int64]
    L.geo_jrand_next_int.argtypes = [C.c_void_p]; L.geo_jrand_next_int.restype = C.c_int32
    L.geo_jrand_next_int_bound.argtypes = [C.c_void_p, C.c_int32]; L.geo_jrand_next_int_bound.restype = C.c_int32
    L.geo_jrand_next_float.argtypes = [C.c_void_p]; L.geo_jrand_next_float.restype = C.c_float
    L.geo_jrand_shuffle.argtypes = [C.c_void_p, i32p, C.c_int32]
    L.geo_bca_build.argtypes = [C.c_int32, i64p, i32p, f32p, i64p, i32p, f32p,
                                C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p]
    L.geo_bca_build.restype = C.c_int
    L.geo_coo_free.argtypes = [C.c_void_p]
    L.geo_bca_single.argtypes = [C.c_int32, i64p, i32p, f32p, i64p, i32p, f32p,
                                 C.c_double, C.c_double, C.c_int, C.c_int, C.c_int32,
                                 i32p, f32p, C.c_int64]
    L.geo_bca_single.restype = C.c_int64
    L.geo_hashmap_replay.argtypes = [C.c_int64, i32p, i32p, i32p, C.c_int64, i32p, i32p]
    L.geo_hashmap_replay.restype = C.c_int64
    L.geo_glove_create.argtypes = [C.c_int32, C.c_int32, C.c_int64, i32p, i32p, f32p,
                                   C.c_double, C.c_int, C.c_int64, C.c_int]
    L.geo_glove_create.restype = C.c_void_p
    L.geo_glove_create_opt.argtypes = [C.c_int32, C.c_int32, C.c_int64, i32p, i32p, f32p,
                                       C.c_double, C.c_int, C.c_int64, C.c_int, C.c_int]
    L.geo_glove_create_opt.restype = C.c_void_p
    L.geo_glove_set_iteration.argtypes = [C.c_void_p, C.c_int]
    L.geo_opt_job.argtypes = [C.c_int, C.c_int, C.c_int32, C.c_int64, i32p, i32p, f32p, C.c_double, C.c_int] + [f32p] * 12
    L.geo_opt_job.restype = C.c_float
    L.geo_glove_destroy.argtypes = [C.c_void_p]
    L.geo_glove_epoch.argtypes = [C.c_void_p, C.c_int]; L.geo_glove_epoch.restype = C.c_double
    L.geo_glove_epoch_noshuffle.argtypes = [C.c_void_p, C.c_int]; L.geo_glove_epoch_noshuffle.restype = C.c_double
    L.geo_glove_optimize.argtypes = [C.c_void_p, C.c_int, C.c_double, f64p, f64p, C.c_int]
    L.geo_glove_optimize.restype = C.c_int
    L.geo_glove_extract.argtypes = [C.c_void_p, f64p]
    for nm in STATE_NAMES:
        f = getattr(L, "geo_glove_" + nm); f.argtypes = [C.c_void_p]; f.restype = f32p
    L.geo_glove_perm.argtypes = [C.c_void_p]; L.geo_glove_perm.restype = i32p
    L.geo_glove_rng_state.argtypes = [C.c_void_p]; L.geo_glove_rng_state.restype = C.c_uint64
    L.geo_adagrad_job.argtypes = [C.c_int32, C.c_int64, i32p, i32p, f32p, C.c_double, C.c_int,
                                  f32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p]
    L.geo_adagrad_job.restype = C.c_float
    L.geo_format_11_6E.argtypes = [C.c_double, C.c_char_p, C.c_int]; L.geo_format_11_6E.restype = C.c_int
    u16p = C.POINTER(C.c_uint16)
    L.geo_sim_pair.argtypes = [C.POINTER(SimCfg), u16p, C.c_int32, u16p, C.c_int32, C.POINTER(C.c_int)]; L.geo_sim_pair.restype = C.c_double
    L.geo_sim_jarowinkler.argtypes = [u16p, C.c_int32, u16p, C.c_int32]; L.geo_sim_jarowinkler.restype = C.c_double
    L.geo_sim_levenshtein_distance.argtypes = [u16p, C.c_int32, u16p, C.c_int32]; L.geo_sim_levenshtein_distance.restype = C.c_int32
    L.geo_sim_pattern_supported.argtypes = [C.c_char_p]; L.geo_sim_pattern_supported.restype = C.c_int
    L.geo_compare_group.argtypes = [C.POINTER(SimCfg), i64p, u16p, i32p, i32p, C.c_int32, i32p, i32p, C.c_int32, C.c_int,
                                    i32p, i32p, f32p, C.c_int64]
    L.geo_compare_group.restype = C.c_int64
    _lib = L
    return L


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class JavaRandom:
    """java.util.Random / ExtendedRandom restatement."""

    def __init__(self, seed):
        self._s = C.c_uint64(0)
        lib().geo_jrand_init(C.byref(self._s), seed)

    @property
    def state(self):
        return self._s.value

    def next_int(self, bound=None):
        if bound is None:
            return lib().geo_jrand_next_int(C.byref(self._s))
        return lib().geo_jrand_next_int_bound(C.byref(self._s), bound)

    def next_float(self):
        return lib().geo_jrand_next_float(C.byref(self._s))

    def shuffle(self, a):
        assert a.dtype == np.int32 and a.flags.c_contiguous
        lib().geo_jrand_shuffle(C.byref(self._s), _p(a, C.c_int32), len(a))


class _Coo(C.Structure):
    _fields_ = [("nnz", C.c_int64), ("I", C.POINTER(C.c_int32)), ("J", C.POINTER(C.c_int32)),
                ("X", C.POINTER(C.c_float)), ("row_ptr", C.POINTER(C.c_int64)),
                ("max", C.c_double), ("V", C.c_int32)]


def _csr_args(csr):
    ptr, idx, w = csr
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    return ptr, idx, w


def bca_build(V, out_csr, in_csr, alpha, epsilon, directed=True, normalize=NORM_NONE):
    """BookmarkColoring ctor. Returns dict(I, J, X, row_ptr, max, nnz)."""
    op, oi, ow = _csr_args(out_csr)
    ip, ii, iw = _csr_args(in_csr)
    coo = _Coo()
    rc = lib().geo_bca_build(V, _p(op, C.c_int64), _p(oi, C.c_int32), _p(ow, C.c_float),
                             _p(ip, C.c_int64), _p(ii, C.c_int32), _p(iw, C.c_float),
                             alpha, epsilon, int(directed), normalize, C.byref(coo))
    if rc != 0:
        raise RuntimeError("geo_bca_build failed")
    n = coo.nnz
    out = dict(
        I=np.ctypeslib.as_array(coo.I, shape=(max(n, 1),))[:n].copy(),
        J=np.ctypeslib.as_array(coo.J, shape=(max(n, 1),))[:n].copy(),
        X=np.ctypeslib.as_array(coo.X, shape=(max(n, 1),))[:n].copy(),
        row_ptr=np.ctypeslib.as_array(coo.row_ptr, shape=(V + 1,)).copy(),
        max=coo.max, nnz=n)
    lib().geo_coo_free(C.byref(coo))
    return out


def bca_single(V, out_csr, in_csr, alpha, epsilon, bookmark, directed=True, normalize=NORM_NONE, cap=1 << 16):
    op, oi, ow = _csr_args(out_csr)
    ip, ii, iw = _csr_args(in_csr)
    keys = np.zeros(cap, np.int32); vals = np.zeros(cap, np.float32)
    n = lib().geo_bca_single(V, _p(op, C.c_int64), _p(oi, C.c_int32), _p(ow, C.c_float),
                             _p(ip, C.c_int64), _p(ii, C.c_int32), _p(iw, C.c_float),
                             alpha, epsilon, int(directed), normalize, bookmark,
                             _p(keys, C.c_int32), _p(vals, C.c_float), cap)
    assert 0 <= n <= cap
    return keys[:n].copy(), vals[:n].copy()


def hashmap_replay(ops):
    """ops: [(op, key)] with op 'put' (BCV.add), 'merge' or 'remove'.  Returns (keys in iteration order, table length, tree bins)."""
    code = {"put": 0, "merge": 1, "remove": 2}
    o = np.array([code[a] for a, _ in ops], np.int32); k = np.array([b for _, b in ops], np.int32)
    out = np.zeros(max(len(ops), 1), np.int32)
    tl = C.c_int32(0); tb = C.c_int32(0)
    n = lib().geo_hashmap_replay(len(ops), _p(o, C.c_int32), _p(k, C.c_int32), _p(out, C.c_int32), len(out), C.byref(tl), C.byref(tb))
    return out[:n].tolist(), tl.value, tb.value


class Glove:
    """Adagrad optimizer restatement (Optimizer + Adagrad + Glove/PGloveCost)."""

    def __init__(self, V, D, I, J, X, xmax, cost=COST_GLOVE, seed=42, threads=1, opt=OPT_ADAGRAD):
        I = np.ascontiguousarray(I, np.int32); J = np.ascontiguousarray(J, np.int32)
        X = np.ascontiguousarray(X, np.float32)
        self.V, self.D, self.N = V, D, len(I)
        self.opt = opt
        self._h = lib().geo_glove_create_opt(V, D, self.N, _p(I, C.c_int32), _p(J, C.c_int32), _p(X, C.c_float),
                                             float(xmax), cost, seed, threads, opt)
        if not self._h:
            raise MemoryError

    def close(self):
        if self._h:
            lib().geo_glove_destroy(self._h); self._h = None

    def __del__(self):
        self.close()

    def epoch(self, race=False, shuffle=True):
        f = lib().geo_glove_epoch if shuffle else lib().geo_glove_epoch_noshuffle
        return f(self._h, int(race))

    def optimize(self, maxiter, tolerance, race=False):
        hist = np.zeros(maxiter, np.float64); fin = C.c_double(0)
        n = lib().geo_glove_optimize(self._h, maxiter, tolerance, _p(hist, C.c_double), C.byref(fin), int(race))
        return hist[:n].copy(), fin.value

    def extract(self):
        out = np.zeros(self.V * self.D, np.float64)
        lib().geo_glove_extract(self._h, _p(out, C.c_double))
        return out.reshape(self.V, self.D)

    def _arr(self, name, n):
        p = getattr(lib(), "geo_glove_" + name)(self._h)
        return np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n]

    # live views into the oracle's state
    @property
    def focus(self): return self._arr("focus", self.V * self.D).reshape(self.V, self.D)
    @property
    def context(self): return self._arr("context", self.V * self.D).reshape(self.V, self.D)
    @property
    def fbias(self): return self._arr("fbias", self.V)
    @property
    def cbias(self): return self._arr("cbias", self.V)
    @property
    def gsq_focus(self): return self._arr("gsq_focus", self.V * self.D).reshape(self.V, self.D)
    @property
    def gsq_context(self): return self._arr("gsq_context", self.V * self.D).reshape(self.V, self.D)
    @property
    def gsq_fbias(self): return self._arr("gsq_fbias", self.V)
    @property
    def gsq_cbias(self): return self._arr("gsq_cbias", self.V)
    @property
    def m2_focus(self): return self._arr("m2_focus", self.V * self.D).reshape(self.V, self.D)
    @property
    def m2_context(self): return self._arr("m2_context", self.V * self.D).reshape(self.V, self.D)
    @property
    def m2_fbias(self): return self._arr("m2_fbias", self.V)
    @property
    def m2_cbias(self): return self._arr("m2_cbias", self.V)
    @property
    def perm(self): return self._arr("perm", self.N)
    @property
    def rng_state(self): return lib().geo_glove_rng_state(self._h)

    def state(self, full=None):
        """Copy of the tables; the m2_* tables are included for Adam/AMSGrad (or with full=True)."""
        names = STATE_NAMES if (full if full is not None else self.opt != OPT_ADAGRAD) else STATE_NAMES[:8]
        return {n: getattr(self, n).copy() for n in names}

    def set_iteration(self, it):
        lib().geo_glove_set_iteration(self._h, it)


def adagrad_job(D, I, J, X, xmax, cost, state):
    """Bare Adagrad.createJob loop over (I,J,X) in the given order on a state dict (in place)."""
    I = np.ascontiguousarray(I, np.int32); J = np.ascontiguousarray(J, np.int32)
    X = np.ascontiguousarray(X, np.float32)
    names = ("focus", "context", "fbias", "cbias", "gsq_focus", "gsq_context", "gsq_fbias", "gsq_cbias")
    for k in names:
        assert state[k].dtype == np.float32 and state[k].flags.c_contiguous
    return lib().geo_adagrad_job(D, len(I), _p(I, C.c_int32), _p(J, C.c_int32), _p(X, C.c_float),
                                 float(xmax), cost, *[_p(state[k], C.c_float) for k in names])


def opt_job(opt, iteration, D, I, J, X, xmax, cost, state):
    """One job of Adagrad / Adam / AMSGrad over (I,J,X) in the given order on a 12-table state dict (in place)."""
    I = np.ascontiguousarray(I, np.int32); J = np.ascontiguousarray(J, np.int32)
    X = np.ascontiguousarray(X, np.float32)
    for k in STATE_NAMES:
        assert state[k].dtype == np.float32 and state[k].flags.c_contiguous, k
    return lib().geo_opt_job(opt, iteration, D, len(I), _p(I, C.c_int32), _p(J, C.c_int32), _p(X, C.c_float),
                             float(xmax), cost, *[_p(state[k], C.c_float) for k in STATE_NAMES])


def format_11_6E(v):
    buf = C.create_string_buffer(64)
    lib().geo_format_11_6E(float(v), buf, 64)
    return buf.value.decode()
