"""Engine._check_device_input without a GPU: fake device tensors (the check never dereferences the pointer)."""
import numpy as np
import pytest


class FakeDev:
    def __init__(self, index):
        self.index = index


class FakeTensor:
    is_cuda = True

    def __init__(self, shape, dtype="float32", device=0, contiguous=True, cuda=True):
        self.shape, self.dtype, self.device = tuple(shape), f"torch.{dtype}", FakeDev(device)
        self._c, self.is_cuda = contiguous, cuda

    def is_contiguous(self):
        return self._c

    def numel(self):
        return int(np.prod(self.shape))

    def element_size(self):
        return {"float32": 4, "float64": 8, "int32": 4}[self.dtype[6:]]

    def data_ptr(self):
        return 0x1000


def _engine(dtype="f32", device=0):
    from rcognita_amd import Engine, EngineConfig

    e = Engine.__new__(Engine)  # no handle: only the host-side checks are exercised
    e.cfg = EngineConfig(sys_id=0, batch=8, dtype=dtype, device=device)
    e._h = None
    e.B, e.N, e.du, e.ds, e.dy, e.dc = 8, 5, 2, 5, 5, 7
    e.real = np.float32 if dtype == "f32" else np.float64
    return e


def test_accepts_the_exact_layout_only():
    e = _engine()
    ptr, K = e._cand(FakeTensor((8, 64, 5, 2)), [])
    assert K == 64 and ptr.value == 0x1000
    for bad, msg in ((FakeTensor((8, 64, 5, 2), "float64"), "dtype"), (FakeTensor((7, 64, 5, 2)), "shape"),
                     (FakeTensor((8, 64, 6, 2)), "shape"), (FakeTensor((8, 64, 5, 2), device=1), "cuda:1"),
                     (FakeTensor((8, 64, 5, 2), contiguous=False), "contiguous"),
                     (FakeTensor((8, 64, 5, 2), cuda=False), "CUDA"), (FakeTensor((8, 64, 10)), r"\[B, K, N, du\]")):
        with pytest.raises(ValueError, match=msg):
            e._cand(bad, [])
    with pytest.raises(ValueError, match="K = 32"):
        e._cand(FakeTensor((8, 64, 5, 2)), [], K=32)
    assert e._cand(None, [], K=49) == (None, 49)
    assert e._in_soa(FakeTensor((5, 8)), [], 5, "obs").value == 0x1000
    with pytest.raises(ValueError, match="shape"):
        e._in_soa(FakeTensor((8, 5)), [], 5, "obs")  # host layout handed over as a device tensor
    e64 = _engine("f64", device=1)
    assert e64._cand(FakeTensor((8, 16, 5, 2), "float64", device=1), [])[1] == 16
    with pytest.raises(ValueError, match="dtype"):
        e64._cand(FakeTensor((8, 16, 5, 2), "float32", device=1), [])
