"""The operator surface accepts device arrays from producers other than torch -- a DLPack capsule, an object with
``__dlpack__`` / ``__dlpack_device__``, an object with ``__cuda_array_interface__`` over memory torch never
allocated -- zero-copy, with nanobind's ``.noconvert()`` behaviour kept (reference typing:
gpu_rf_gen/xgpr_cuda_rfgen_cpp_ext.cpp:20-93, ``nb::ndarray<..., nb::device::cuda>``)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


class DLPackOnly:
    """A producer that is not a torch tensor: exposes nothing but the DLPack protocol."""

    def __init__(self, t):
        self._t = t

    def __dlpack__(self, stream=None):
        return self._t.__dlpack__(stream=stream)

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()


class HipBuffer:
    """Device memory from hipMalloc (libamdhip64 through ctypes), exported through __cuda_array_interface__."""

    def __init__(self, host):
        self.hip = C.CDLL("libamdhip64.so")
        self.host = np.ascontiguousarray(host)
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(self.host.nbytes)) == 0
        assert self.hip.hipMemcpy(self.ptr, C.c_void_p(self.host.ctypes.data), C.c_size_t(self.host.nbytes), 1) == 0
        self.__cuda_array_interface__ = {"shape": self.host.shape, "typestr": self.host.dtype.str,
                                         "data": (self.ptr.value, False), "version": 3, "strides": None}

    def download(self):
        out = np.empty_like(self.host)
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(C.c_void_p(out.ctypes.data), self.ptr, C.c_size_t(out.nbytes), 2) == 0
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def test_fht_in_place_through_every_protocol():
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    g = load_golden("g1_fht.npz")
    x, want = g["x_1024"], g["y32_1024"]
    # raw DLPack capsule
    t = torch.from_numpy(x.copy()).to(DEV)
    ext.cudaFastHadamardTransform2D(torch.utils.dlpack.to_dlpack(t))
    assert np.array_equal(t.cpu().numpy(), want)
    # object with __dlpack__ only
    t = torch.from_numpy(x.copy()).to(DEV)
    ext.cudaFastHadamardTransform2D(DLPackOnly(t))
    assert np.array_equal(t.cpu().numpy(), want)
    # __cuda_array_interface__ over memory torch did not allocate: the operator must write THAT memory
    buf = HipBuffer(x.copy())
    try:
        ext.cudaFastHadamardTransform2D(buf)
        torch.cuda.synchronize()
        assert np.array_equal(buf.download(), want)
    finally:
        buf.free()


def test_feature_gen_with_mixed_producers():
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    g = load_golden("g2_rbf.npz")
    si = 0
    x, radem, chi = g[f"x_{si}"], g[f"radem_{si}"], g[f"chi_{si}"]
    want = g[f"out32_{si}"]
    scale = np.sqrt(1.0 / chi.shape[0])
    out = HipBuffer(np.zeros_like(want))
    try:
        ext.cudaRBFFeatureGen(DLPackOnly(torch.from_numpy(x.copy()).to(DEV)), out,
                              torch.utils.dlpack.to_dlpack(torch.from_numpy(radem).to(DEV)),
                              torch.from_numpy(chi).to(DEV), bool(g[f"intercept_{si}"]))
        torch.cuda.synchronize()
        assert np.abs(out.download() - want).max() <= 4e-7 * scale
    finally:
        out.free()


def test_noconvert_semantics_kept():
    from xgpr_amd import xgpr_hip_rfgen_ext as ext
    with pytest.raises(TypeError):                     # host array (numpy speaks DLPack too): not copied to the device
        ext.cudaFastHadamardTransform2D(np.zeros((4, 8), np.float32))
    with pytest.raises(TypeError):                     # not an array at all
        ext.cudaFastHadamardTransform2D([[1.0, 2.0]])
    with pytest.raises(TypeError):                     # wrong dtype through DLPack: refused, not cast
        ext.cudaSRHT(DLPackOnly(torch.zeros((4, 8), device=DEV)), DLPackOnly(torch.ones(8, dtype=torch.int32, device=DEV)))
    with pytest.raises(TypeError):                     # non-contiguous view through DLPack
        ext.cudaFastHadamardTransform2D(DLPackOnly(torch.zeros((8, 8), device=DEV).T))
    with pytest.raises(TypeError):                     # half precision has no overload
        ext.cudaFastHadamardTransform2D(DLPackOnly(torch.zeros((4, 8), dtype=torch.float16, device=DEV)))
