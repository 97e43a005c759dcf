"""The Python stand-in with the lazy binding on (NDArray.set_lazy_binding / NP_LAZY_BINDING=1): it then behaves like a `--with-hip` tree
with INTEGRATION.md 2c applied — operators and unary methods append through ext/hip_lazy.c's appenders, `_p` is the marshalling point
(buffer_get) where pending values are computed.  `NP_LAZY_BINDING=1 python -m pytest tests -m gpu` runs the WHOLE GPU suite that way
(profiles/r06/pytest_lazy_binding.log: the same 1228 tests pass); here: that the switch really defers and fuses, and what it costs."""
import ctypes as C

import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


def _launches():
    from numpower_amd._lib import load
    n = C.c_ulonglong(0)
    assert load().np_debug_launch_count(C.byref(n)) == 0
    return n.value


def test_operators_defer_and_fuse_under_the_lazy_binding(hip):
    from numpower_amd import ndarray as nd
    from numpower_amd.ndarray import NDArray, _load_host
    h = _load_host()
    x = synth.uniform((300, 257), 5, -2.0, 2.0)
    y = synth.uniform((300, 257), 6, 0.5, 3.0)
    gx, gy = NDArray.array(x).gpu(), NDArray.array(y).gpu()
    eager = (NDArray.exp(gx) * gy + 2.0).cpu().toArray()
    eager_sum = NDArray.sum(NDArray.exp(gx) * gy)
    was = nd._LAZY_BINDING
    NDArray.set_lazy_binding(True)
    try:
        l0 = _launches()
        c = NDArray.exp(gx) * gy + 2.0                      # three PHP-level ops
        assert _launches() == l0 and h.NPH_IsPending(c._ptr) == 1 and h.NPH_PendingCount() == 1      # nothing ran; the temporaries are gone
        assert c.shape() == [300, 257]                       # a consumer (shape() marshals like every method): the value is computed
        assert _launches() == l0 + 1 and h.NPH_IsPending(c._ptr) == 0
        lazy = c.cpu().toArray()
        assert (np.asarray(lazy, dtype=np.float32).view(np.uint32) == np.asarray(eager, dtype=np.float32).view(np.uint32)).all()
        # a full reduction of a pending value: inside the chain's kernel, one launch, the operand stays pending
        l1 = _launches()
        d = NDArray.exp(gx) * gy
        s = NDArray.sum(d)
        assert _launches() == l1 + 1 and h.NPH_IsPending(d._ptr) == 1
        assert abs(s - eager_sum) <= 2e-6 * abs(eager_sum)
        del d                                                # dropped unevaluated
        assert h.NPH_PendingCount() == 0
        # a write to an input after the expression was built: the expression saw the old values
        e = gx + 1.0
        gx.fill(0.0)
        assert (np.asarray(e.cpu().toArray(), dtype=np.float32) == x + np.float32(1.0)).all()
        assert (np.asarray(gx.cpu().toArray()) == 0).all()
    finally:
        NDArray.set_lazy_binding(was)
