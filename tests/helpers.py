"""Shared helpers for the parity tests."""
import numpy as np

import geglove
from geglove import capi
import oracle as O


def make_config(dim, method="glove", threads=1, maxiter=5, tolerance=0.0, opt="adagrad", **device):
    return geglove.Configuration({
        "graph": "synthetic", "method": method, "dim": dim, "threads": threads,
        "bca": {"alpha": 0.1, "epsilon": 1e-3, "directed": True},
        "opt": {"method": opt, "tolerance": tolerance, "maxiter": maxiter},
        "output": {"uri": []}, "device": device})


def cost_kind(method):
    return O.COST_GLOVE if method == "glove" else O.COST_PGLOVE


OPT_KIND = {"adagrad": O.OPT_ADAGRAD, "adam": O.OPT_ADAM, "amsgrad": O.OPT_AMSGRAD}


def assert_state_equal(dev_state, ora, exact=True, rtol=0.0, atol=0.0, what=""):
    ora_state = ora.state() if hasattr(ora, "state") else ora
    for name in dev_state:
        a = np.asarray(dev_state[name]).reshape(-1)
        b = np.asarray(ora_state[name]).reshape(-1)
        assert a.shape == b.shape, (name, a.shape, b.shape)
        if exact:
            bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
            assert bad.size == 0, "%s %s: %d of %d words differ, first at %d: %r vs %r" % (
                what, name, bad.size, a.size, bad[0], a[bad[0]], b[bad[0]])
        else:
            # atol is relative to the table's largest magnitude (elements near zero carry the
            # absolute rounding error of the update, not a relative one)
            scale = float(np.max(np.abs(b))) if b.size else 1.0
            np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale, err_msg="%s %s" % (what, name))
