"""numpy model of the ARITHMETIC of k_adagrad_runs' Adam / AMSGrad instances (csrc/glove.hip: moment_step, the per-lane fma
chain of the dot product, wave_sum's reduction tree), bit for bit: fp32 throughout, explicit fused multiply-adds where the kernel
calls __builtin_fmaf, correctly rounded sqrt and reciprocal (without fast-math __fsqrt_rn / __frcp_rn are __ocml_sqrt_rte_f32 /
__ocml_div_rte_f32).  A one-worker run of the kernel is a sequential program; replaying its order (ge_glove_epoch_order) through
this model must give the same tables up to what the two `log` implementations may disagree on (one fp64 ulp, before a narrowing
to fp32).  The fp64 oracle (Adam.java's own arithmetic) stays the loose reference; this is the tight one.
Test infrastructure only."""
import numpy as np

F32, F64 = np.float32, np.float64
BETA1, BETA2, EPS = F32(0.9), F32(0.999), F32(1e-7)
OMB1, OMB2 = F32(1) - BETA1, F32(1) - BETA2          # evaluated in fp32, as `1 - BETA1` is in the kernel (and in Adam.java:45-53)


def fma32(a, b, c):
    """round_to_fp32(a * b + c) with ONE rounding, elementwise on fp32 arrays (v_fma_f32).  The product of two fp32 values is exact
    in fp64; the sum is rounded to fp64 and then to fp32, which double-rounds only when the fp64 sum lands exactly on the midpoint
    of two fp32 neighbours while the discarded part is non-zero -- detected with the exact error of the fp64 addition (TwoSum)."""
    a, b, c = (np.asarray(x, F32) for x in (a, b, c))
    p = a.astype(F64) * b.astype(F64)
    c64 = c.astype(F64)
    s = p + c64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)                       # exact: p + c == s + err
    r = s.astype(F32)
    d = s - r.astype(F64)
    away = np.where(d > 0, F32(np.inf), F32(-np.inf)).astype(F32)
    n = np.nextafter(r, away)
    with np.errstate(invalid="ignore", over="ignore"):
        tie = (d != 0) & (s == (r.astype(F64) + n.astype(F64)) * 0.5) & (err != 0) & np.isfinite(s)
    if np.any(tie):
        hi, lo = np.maximum(r, n), np.minimum(r, n)
        r = np.where(tie, np.where(err > 0, hi, lo), r).astype(F32)
    return r


_L = np.arange(64)
_DPP = (_L ^ 1, _L ^ 2, (_L & ~7) | (7 - (_L & 7)), (_L & ~15) | (15 - (_L & 15)))     # quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror


def wave_sum(v):
    """glove.hip wave_sum: four DPP adds inside each row of 16 lanes, then (r0 + r16) + (r32 + r48)."""
    v = np.asarray(v, F32)
    for perm in _DPP:
        v = (v + v[perm]).astype(F32)
    return F32(F32(v[0] + v[16]) + F32(v[32] + v[48]))


def lane_dot(a, b, vw, nch):
    """The kernel's dot product of two rows of D floats: lane L holds elements [vw * (L + 64 q), + vw) of register chunk q and runs
    one fma chain over q, then t; lanes past the row contribute 0; then wave_sum."""
    D = a.shape[0]
    pad = nch * 64 * vw
    A = np.zeros(pad, F32); B = np.zeros(pad, F32)
    A[:D] = a; B[:D] = b
    A = A.reshape(nch, 64, vw); B = B.reshape(nch, 64, vw)
    part = np.zeros(64, F32)
    for q in range(nch):
        inr = (np.arange(64) + q * 64) * vw < D
        for t in range(vw):
            part = np.where(inr, fma32(A[q, :, t], B[q, :, t], part), part)
    return wave_sum(part)


def moment_step(amsgrad, corr, grad, m, v):
    """glove.hip moment_step, elementwise: returns (step, m', v')."""
    grad = np.asarray(grad, F32)
    m2 = fma32(BETA1, m, (OMB1 * grad).astype(F32))
    vn = fma32(BETA2, v, (OMB2 * (grad * grad).astype(F32)).astype(F32))
    v2 = np.maximum(v, vn).astype(F32) if amsgrad else vn
    den = (np.sqrt(v2).astype(F32) + EPS).astype(F32)
    step = ((F32(corr) * m2).astype(F32) * (F32(1) / den).astype(F32)).astype(F32)
    return step, m2, v2


def cost_terms(kind_pglove, x, xmax):
    """ge_cost.h cost_terms<false>: (l fp64, w fp32) of one nonzero."""
    x = F32(x)
    if kind_pglove:
        return F64(np.log(F64(F32(x / F32(F32(1) - x))))), x
    r = F64(x) / F64(xmax)
    q = np.sqrt(r)
    w = F32(1) if F64(x) > F64(xmax) else F32(q * np.sqrt(q))
    return F64(np.log(F64(x))), w


def adam_correction(lr, iteration):
    """glove.hip fill_params (Adam.java:84): fp64 from the fp32 constants, narrowed where the kernel uses it."""
    lr, b1, b2 = F64(F32(lr)), F64(F32(0.9)), F64(F32(0.999))
    return F32(lr * np.sqrt(1 - b2 ** F64(iteration + 1)) / (1 - b1 ** F64(iteration + 1)))


def moment_epoch(amsgrad, iteration, D, vw, nch, I, J, X, xmax, state, lr=0.05, pglove=False):
    """One sequential pass over (I, J, X) in the given order on the 12-table state dict (2-D row tables, in place).  Returns the
    cost sum (fp64, accumulated in walk order)."""
    corr = F32(lr) if amsgrad else adam_correction(lr, iteration)
    foc, ctx = state["focus"], state["context"]
    m1f, m1c, m2f, m2c = state["gsq_focus"], state["gsq_context"], state["m2_focus"], state["m2_context"]
    fb, cb = state["fbias"], state["cbias"]
    m1fb, m1cb, m2fb, m2cb = state["gsq_fbias"], state["gsq_cbias"], state["m2_fbias"], state["m2_cbias"]
    cost = 0.0
    for i, j, x in zip(I.tolist(), J.tolist(), X.tolist()):
        l, w = cost_terms(pglove, x, xmax)
        a, b = foc[i].copy(), ctx[j].copy()
        dot = lane_dot(a, b, vw, nch)
        ic = F32(F64(dot) + (F64(F32(fb[i] + cb[j])) - l))
        wc = F32(w * ic)
        cost += (0.5 * float(wc)) * float(ic)
        st, m1c[j], m2c[j] = moment_step(amsgrad, corr, (wc * a).astype(F32), m1c[j], m2c[j])
        ctx[j] = (b - st).astype(F32)
        st, m1f[i], m2f[i] = moment_step(amsgrad, corr, (wc * b).astype(F32), m1f[i], m2f[i])
        foc[i] = (a - st).astype(F32)
        st, m, v = moment_step(amsgrad, corr, wc, m1fb[i], m2fb[i]); fb[i] = F32(fb[i] - st); m1fb[i] = m; m2fb[i] = v
        st, m, v = moment_step(amsgrad, corr, wc, m1cb[j], m2cb[j]); cb[j] = F32(cb[j] - st); m1cb[j] = m; m2cb[j] = v
    return cost
