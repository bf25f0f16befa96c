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
_LIB_PATH = os.path.join(_HERE, "libge_oracle.so")

NORM_NONE, NORM_UNITY, NORM_COUNTS = 0, 1, 2
COST_GLOVE, COST_PGLOVE = 0, 1
OPT_ADAGRAD, OPT_ADAM, OPT_AMSGRAD = 0, 1, 2
STATE_NAMES = ("focus", "context", "fbias", "cbias", "gsq_focus", "gsq_context", "gsq_fbias", "gsq_cbias",
               "m2_focus", "m2_context", "m2_fbias", "m2_cbias")


def build(force=False):
    src = os.path.join(_HERE, "ge_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
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
