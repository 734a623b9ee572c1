"""Operator surface of the reference's GPU extension module, served by libxgpr_hip.so.

Mirrors ``xGPR.xgpr_cuda_rfgen_cpp_ext`` (reference:
src/xGPR/random_feature_generation/gpu_rf_gen/xgpr_cuda_rfgen_cpp_ext.cpp:20-93): same
function names (``cuda*`` kept as aliases of the ``hip*`` names), same keyword names, same
in-place semantics, same error behaviour -- arguments are *not* converted (wrong dtype,
device or contiguity raises ``TypeError`` like nanobind's ``.noconvert()``), validation
failures raise ``RuntimeError`` with the reference's messages, and every function returns 0.

Arrays are device arrays of any producer that speaks DLPack (``__dlpack__`` / a DLPack capsule; a torch-ROCm
tensor exports device type kDLROCM) or ``__cuda_array_interface__`` -- what nanobind's
``nb::ndarray<..., nb::device::cuda>`` accepts in the reference, so its cupy callers
(kernels/basic_kernels/sorf_kernel_baseclass.py:104-126) can pass their arrays as they are.  They are adopted
zero-copy (``torch.from_dlpack`` / ``torch.as_tensor``): the in-place operators write the producer's own memory.
``seqlengths`` stays on the HOST (int32 numpy array or CPU tensor), as in the reference's CUDA module
(gpu_rf_gen/convolution_ops/rbf_convolution.h:19).  Calls are asynchronous on torch's current stream.
"""
import ctypes as C
import functools
import inspect

import numpy as np
import torch

from . import _lib

_LIB = _lib.load()

_T2S = {torch.float32: "f32", torch.float64: "f64"}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _adopt(obj, name):
    """A zero-copy torch view of a device array from any producer; never a conversion (a host array stays a host
    array and is then refused by _dev, like nanobind's .noconvert())."""
    if isinstance(obj, torch.Tensor) or obj is None:
        return obj
    try:
        if hasattr(obj, "__dlpack__") or type(obj).__name__ == "PyCapsule":
            return torch.from_dlpack(obj)
        if hasattr(obj, "__cuda_array_interface__"):
            return torch.as_tensor(obj, device="cuda")
    except (RuntimeError, ValueError, BufferError) as exc:
        raise TypeError(f"{name}: could not adopt the array through DLPack / __cuda_array_interface__: {exc}") from exc
    raise TypeError(f"{name}: expected a device array (torch tensor, DLPack or __cuda_array_interface__ producer)")


def _array_args(*names):
    """Decorator: the named arguments may come from any DLPack / __cuda_array_interface__ producer."""
    def wrap(fn):
        sig = inspect.signature(fn)

        @functools.wraps(fn)
        def inner(*args, **kwargs):
            bound = sig.bind(*args, **kwargs)
            for nm in names:
                if nm in bound.arguments:
                    bound.arguments[nm] = _adopt(bound.arguments[nm], nm)
            return fn(*bound.args, **bound.kwargs)
        return inner
    return wrap


def _dev(t, name, dtype, ndim):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a device array")
    if not t.is_cuda:
        raise TypeError(f"{name}: expected a device tensor, got a host tensor")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.dim() != ndim:
        raise TypeError(f"{name}: expected {ndim} dims, got {t.dim()}")
    if not t.is_contiguous():
        raise TypeError(f"{name}: expected a C-contiguous array")
    return C.c_void_p(t.data_ptr())


def _ftype(t, name):
    if not isinstance(t, torch.Tensor) or t.dtype not in _T2S:
        raise TypeError(f"{name}: expected a float32 or float64 torch tensor")
    return _T2S[t.dtype]


def _workspace(nbytes, device):
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
    return ws, C.c_void_p(ws.data_ptr()), C.c_size_t(ws.numel())


def _sorf_ws(radem, width, inputArr):
    nbytes = _LIB.xgpr_sorf_workspace_bytes(radem.shape[2], int(width), inputArr.element_size())
    return _workspace(nbytes, inputArr.device)


def _conv_ws(radem, width, inputArr):
    """Workspace of a convolution operator call, with room for the longest-first processing order."""
    nbytes = _LIB.xgpr_conv_workspace_bytes(radem.shape[2], int(width), inputArr.element_size(), inputArr.shape[0])
    return _workspace(nbytes, inputArr.device)


def _seqlens(seqlengths, device):
    """Host int32 array (validated by the library on the host) + its device copy."""
    if isinstance(seqlengths, torch.Tensor):
        if seqlengths.is_cuda or seqlengths.dtype != torch.int32 or seqlengths.dim() != 1:
            raise TypeError("seqlengths: expected a 1-d int32 array on the host")
        host = seqlengths.contiguous().numpy()
    elif isinstance(seqlengths, np.ndarray):
        if seqlengths.dtype != np.int32 or seqlengths.ndim != 1 or not seqlengths.flags["C_CONTIGUOUS"]:
            raise TypeError("seqlengths: expected a C-contiguous 1-d int32 array on the host")
        host = seqlengths
    else:
        raise TypeError("seqlengths: expected a numpy array or CPU tensor (int32)")
    dev = torch.from_numpy(host).to(device, non_blocking=False)
    return host, dev


@_array_args("inputArr")
def hipFastHadamardTransform2D(inputArr):
    """In-place un-normalised FHT over the last axis of a 2-d array
    (cudaFastHadamardTransform2D, xgpr_cuda_rfgen_cpp_ext.cpp:21-24)."""
    s = _ftype(inputArr, "inputArr")
    p = _dev(inputArr, "inputArr", None, 2)
    return _lib.check(getattr(_LIB, f"xgpr_fht_{s}")(p, inputArr.shape[0], 1, inputArr.shape[1], _stream()))


@_array_args("inputArr")
def hipFastHadamardTransform(inputArr):
    """3-d form (cpuFastHadamardTransform, cpu_rf_gen/xgpr_cpu_rfgen_cpp_ext.cpp:24-30)."""
    s = _ftype(inputArr, "inputArr")
    p = _dev(inputArr, "inputArr", None, 3)
    return _lib.check(getattr(_LIB, f"xgpr_fht_{s}")(p, inputArr.shape[0], inputArr.shape[1],
                                                      inputArr.shape[2], _stream()))


@_array_args("inputArr", "radem")
def hipSRHT(inputArr, radem):
    """cudaSRHT (xgpr_cuda_rfgen_cpp_ext.cpp:25-30)."""
    s = _ftype(inputArr, "inputArr")
    p = _dev(inputArr, "inputArr", None, 2)
    r = _dev(radem, "radem", torch.int8, 1)
    return _lib.check(getattr(_LIB, f"xgpr_srht_{s}")(p, r, inputArr.shape[0], inputArr.shape[1],
                                                       radem.shape[0], _stream()))


def _radem3(radem):
    r = _dev(radem, "radem", torch.int8, 3)
    if radem.shape[0] != 3 or radem.shape[1] != 1:
        raise TypeError("radem: expected shape (3, 1, R)")
    return r


@_array_args("inputArr", "radem", "sampler", "outputArr", "yArr", "ztyOut", "workspace")
def hipSRHTSample(inputArr, radem, sampler, outputArr, ncols=None, yArr=None, ztyOut=None, workspace=None):
    """``outputArr[:, :ncols] = cudaSRHT(pad(inputArr))[:, sampler[:ncols]]`` without touching inputArr
    (srht_compressor.py:87-97 in one pass).  outputArr may have more than ncols columns (row pitch).
    With ``yArr`` the same pass also writes ``inputArr.T @ yArr`` into ``ztyOut``."""
    s = _ftype(inputArr, "inputArr")
    x = _dev(inputArr, "inputArr", None, 2)
    r = _dev(radem, "radem", torch.int8, 1)
    sm = _dev(sampler, "sampler", torch.int64, 1)
    o = _dev(outputArr, "outputArr", inputArr.dtype, 2)
    ncols = sampler.shape[0] if ncols is None else int(ncols)
    if outputArr.shape[0] != inputArr.shape[0] or outputArr.shape[1] < ncols or sampler.shape[0] < ncols:
        raise RuntimeError("incorrect array dims passed")
    yp = zp = wp = C.c_void_p(0)
    wn = C.c_size_t(0)
    if yArr is not None:
        yp = _dev(yArr, "yArr", torch.float64, 1)
        zp = _dev(ztyOut, "ztyOut", torch.float64, 1)
        if yArr.shape[0] != inputArr.shape[0] or ztyOut.shape[0] != inputArr.shape[1]:
            raise RuntimeError("incorrect array dims passed")
        if workspace is None:
            workspace = torch.empty(int(_LIB.xgpr_srht_sample_workspace_bytes(inputArr.shape[1])), dtype=torch.uint8,
                                    device=inputArr.device)
        wp, wn = C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel())
    fn = getattr(_LIB, f"xgpr_srht_sample_{s}")
    return _lib.check(fn(x, r, sm, o, yp, zp, inputArr.shape[0], inputArr.shape[1], radem.shape[0], ncols,
                         outputArr.shape[1], wp, wn, _stream()))


@_array_args("cacheArr", "radem", "sampler", "outputArr", "yArr", "ztyOut", "workspace")
def hipSRHTSampleRows(cacheArr, radem, sampler, outputArr, ncols, fitIntercept, scale=0.0, yArr=None, ztyOut=None,
                      workspace=None):
    """SRHTCompressor.transform_x (srht_compressor.py:87-97) + the chunk's z^T y (rand_nys_constructors.py:115) from
    FLOAT32 feature rows (Z = scale * cacheArr, Z[:, 0] = 1 with fitIntercept; scale = 0: the RBF-family constant):
    ``outputArr[:, :ncols] = cudaSRHT(pad(Z))[:, sampler[:ncols]]`` in float64, remaining columns zeroed.  Any padded
    width up to 32768 (rows beyond the LDS capacity go block by block)."""
    zc = _dev(cacheArr, "cacheArr", torch.float32, 2)
    r = _dev(radem, "radem", torch.int8, 1)
    sm = _dev(sampler, "sampler", torch.int64, 1)
    o = _dev(outputArr, "outputArr", torch.float64, 2)
    ncols = int(ncols)
    if outputArr.shape[0] != cacheArr.shape[0] or outputArr.shape[1] < ncols or sampler.shape[0] < ncols:
        raise RuntimeError("incorrect array dims passed")
    yp = zp = wp = C.c_void_p(0)
    wn = C.c_size_t(0)
    if yArr is not None:
        yp = _dev(yArr, "yArr", torch.float64, 1)
        zp = _dev(ztyOut, "ztyOut", torch.float64, 1)
        if yArr.shape[0] != cacheArr.shape[0] or ztyOut.shape[0] != cacheArr.shape[1]:
            raise RuntimeError("incorrect array dims passed")
        if workspace is None:
            workspace = torch.empty(int(_LIB.xgpr_srht_sample_workspace_bytes(cacheArr.shape[1])), dtype=torch.uint8,
                                    device=cacheArr.device)
        wp, wn = C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel())
    return _lib.check(_LIB.xgpr_srht_sample_rows_f32(zc, r, sm, o, yp, zp, cacheArr.shape[0], cacheArr.shape[1],
                                                     radem.shape[0], ncols, outputArr.shape[1], float(scale),
                                                     int(bool(fitIntercept)), wp, wn, _stream()))


def srht_sample_rows_ok(padded_width, ncols, num_rffs):
    """Whether hipSRHTSampleRows covers this shape (see include/xgpr_hip.h)."""
    nb = max(1, padded_width // 8192)
    return padded_width <= 32768 and (nb == 1 or (nb * ncols <= 8192 and num_rffs % 4 == 0))


@_array_args("aMat", "cacheArr", "outArr", "workspace")
def hipSketchGemm(aMat, cacheArr, outArr, nrows, bt, transOut, fitIntercept, scale=0.0, accumulate=False, workspace=None):
    """``outArr (+)= aMat[:, :nrows].T @ Z`` (bt False: aMat [n, lda], outArr [nrows, num_rffs] or transposed) or
    ``aMat[:, :nrows].T @ Z.T`` (bt True: aMat [num_rffs, lda], outArr [nrows, n] or transposed) on the float64 matrix
    cores, Z = scale * cacheArr float32 rows with Z[:, 0] = 1 when fitIntercept -- the dense products of
    rand_nys_constructors.py:34, :54, :119 without a float64 copy of Z.  aMat's row pitch must be a multiple of 64
    with zeros beyond column nrows."""
    ap = _dev(aMat, "aMat", torch.float64, 2)
    zc = _dev(cacheArr, "cacheArr", torch.float32, 2)
    op = _dev(outArr, "outArr", torch.float64, 2)
    n, m = cacheArr.shape
    nrows = int(nrows)
    jdim, kdim = (n, m) if bt else (m, n)
    if aMat.shape[0] != kdim or aMat.shape[1] < nrows:
        raise RuntimeError("incorrect array dims passed")
    want = (jdim, nrows) if transOut else (nrows, jdim)
    if outArr.shape[0] != want[0] or outArr.shape[1] < want[1]:
        raise RuntimeError("incorrect array dims passed")
    need = int(_LIB.xgpr_sketch_gemm_workspace_bytes(nrows, jdim, kdim, outArr.shape[1], int(bool(transOut))))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=cacheArr.device)
    return _lib.check(_LIB.xgpr_sketch_gemm_f64(ap, aMat.shape[1], zc, n, m, op, outArr.shape[1], nrows, int(bool(bt)),
                                                int(bool(transOut)), float(scale), int(bool(fitIntercept)),
                                                int(bool(accumulate)), C.c_void_p(workspace.data_ptr()),
                                                C.c_size_t(workspace.numel()), _stream()))


def gram_ok(num_rffs, msub):
    """Shapes hipZtZGram covers: whole 128 x 128 tiles of the leading msub features."""
    return msub % 128 == 0 and 0 < msub <= num_rffs and num_rffs % 4 == 0


def hipZtZGram(cacheArr, outArr, fitIntercept, scale=0.0, accumulate=False, workspace=None):
    """``outArr[msub, msub] (+)= Z[:, :msub].T @ Z[:, :msub]`` with msub = outArr.shape[0] on the float64 matrix cores,
    Z = scale * cacheArr float32 rows with Z[:, 0] = 1 when fitIntercept (exact_nmll_calcs.py:42-78, :116-139;
    lb_optimizer.py:68-117) -- both operands from the float32 rows, no float64 copy of Z.  Returns the workspace used
    (pass it back in to avoid reallocation)."""
    zc = _dev(cacheArr, "cacheArr", torch.float32, 2)
    op = _dev(outArr, "outArr", torch.float64, 2)
    n, m = cacheArr.shape
    msub = outArr.shape[0]
    if outArr.shape[1] < msub or not gram_ok(m, msub):
        raise RuntimeError("incorrect array dims passed")
    need = int(_LIB.xgpr_ztz_gram_workspace_bytes(msub, n))
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=cacheArr.device)
    _lib.check(_LIB.xgpr_ztz_gram_f64(zc, n, m, op, outArr.shape[1], msub, float(scale), int(bool(fitIntercept)),
                                      int(bool(accumulate)), C.c_void_p(workspace.data_ptr()),
                                      C.c_size_t(workspace.numel()), _stream()))
    return workspace


def sketch_gemm_workspace_bytes(nrows, jdim, kdim, ldc, trans_out):
    return int(_LIB.xgpr_sketch_gemm_workspace_bytes(nrows, jdim, kdim, ldc, int(bool(trans_out))))


def srht_sample_workspace_bytes(m):
    return int(_LIB.xgpr_srht_sample_workspace_bytes(m))


def srht_sample_ok(padded_width, dtype):
    """Whether hipSRHTSample covers this padded width (the row must fit in LDS)."""
    return padded_width * (8 if dtype == torch.float64 else 4) <= 128 * 1024


@_array_args("inputArr", "outputArr", "radem", "chiArr")
def hipRBFFeatureGen(inputArr, outputArr, radem, chiArr, fitIntercept):
    """cudaRBFFeatureGen (xgpr_cuda_rfgen_cpp_ext.cpp:32-40).  The output is overwritten
    (as by the reference's CUDA kernel, rbf_ops.cu:121-127)."""
    s = _ftype(inputArr, "inputArr")
    x = _dev(inputArr, "inputArr", None, 2)
    o = _dev(outputArr, "outputArr", torch.float64, 2)
    r = _radem3(radem)
    c = _dev(chiArr, "chiArr", inputArr.dtype, 1)
    ws, wp, wn = _sorf_ws(radem, inputArr.shape[1], inputArr)
    return _lib.check(getattr(_LIB, f"xgpr_rbf_feature_gen_{s}")(
        x, o, r, c, inputArr.shape[0], inputArr.shape[1], outputArr.shape[0], outputArr.shape[1],
        chiArr.shape[0], radem.shape[2], int(bool(fitIntercept)), wp, wn, _stream()))


@_array_args("inputArr", "outputArr", "gradArr", "radem", "chiArr")
def hipRBFGrad(inputArr, outputArr, gradArr, radem, chiArr, sigma, fitIntercept):
    """cudaRBFGrad (xgpr_cuda_rfgen_cpp_ext.cpp:41-49)."""
    s = _ftype(inputArr, "inputArr")
    x = _dev(inputArr, "inputArr", None, 2)
    o = _dev(outputArr, "outputArr", torch.float64, 2)
    g = _dev(gradArr, "gradArr", torch.float64, 3)
    if gradArr.shape[2] != 1:
        raise TypeError("gradArr: expected shape (N, M, 1)")
    r = _radem3(radem)
    c = _dev(chiArr, "chiArr", inputArr.dtype, 1)
    ws, wp, wn = _sorf_ws(radem, inputArr.shape[1], inputArr)
    return _lib.check(getattr(_LIB, f"xgpr_rbf_grad_{s}")(
        x, o, g, r, c, inputArr.shape[0], inputArr.shape[1], outputArr.shape[0], outputArr.shape[1],
        gradArr.shape[0], gradArr.shape[1], chiArr.shape[0], radem.shape[2], float(sigma),
        int(bool(fitIntercept)), wp, wn, _stream()))


@_array_args("inputArr", "outputArr", "precompWeights", "sigmaMap", "sigmaVals", "gradArr")
def hipMiniARDGrad(inputArr, outputArr, precompWeights, sigmaMap, sigmaVals, gradArr, fitIntercept):
    """cudaMiniARDGrad (xgpr_cuda_rfgen_cpp_ext.cpp:50-60)."""
    s = _ftype(inputArr, "inputArr")
    x = _dev(inputArr, "inputArr", None, 2)
    o = _dev(outputArr, "outputArr", torch.float64, 2)
    w = _dev(precompWeights, "precompWeights", inputArr.dtype, 2)
    mp = _dev(sigmaMap, "sigmaMap", torch.int32, 1)
    sv = _dev(sigmaVals, "sigmaVals", torch.float64, 1)
    g = _dev(gradArr, "gradArr", torch.float64, 3)
    fn = getattr(_LIB, f"xgpr_mini_ard_grad_{s}")
    return _lib.check(fn(x, o, w, mp, sv, g, inputArr.shape[0], inputArr.shape[1], outputArr.shape[0],
                         outputArr.shape[1], precompWeights.shape[0], precompWeights.shape[1], sigmaMap.shape[0],
                         sigmaVals.shape[0], gradArr.shape[0], gradArr.shape[1], gradArr.shape[2],
                         int(bool(fitIntercept)), _stream()))


@_array_args("inputArr", "outputArr", "radem", "chiArr")
def hipConv1dFGen(inputArr, outputArr, radem, chiArr, seqlengths, convWidth, scalingType):
    """cudaConv1dFGen (xgpr_cuda_rfgen_cpp_ext.cpp:70-80); results are added into outputArr."""
    s = _ftype(inputArr, "inputArr")
    x = _dev(inputArr, "inputArr", None, 3)
    o = _dev(outputArr, "outputArr", torch.float64, 2)
    r = _radem3(radem)
    c = _dev(chiArr, "chiArr", inputArr.dtype, 1)
    host, dev = _seqlens(seqlengths, inputArr.device)
    ws, wp, wn = _conv_ws(radem, int(convWidth) * inputArr.shape[2], inputArr)
    return _lib.check(getattr(_LIB, f"xgpr_conv1d_fgen_{s}")(
        x, o, r, c, C.c_void_p(host.ctypes.data), C.c_void_p(dev.data_ptr()), inputArr.shape[0],
        inputArr.shape[1], inputArr.shape[2], outputArr.shape[0], outputArr.shape[1], chiArr.shape[0],
        radem.shape[2], host.shape[0], int(convWidth), int(scalingType), wp, wn, _stream()))


@_array_args("inputArr", "outputArr", "radem", "chiArr", "gradArr")
def hipConvGrad(inputArr, outputArr, radem, chiArr, seqlengths, gradArr, sigma, convWidth, scalingType):
    """cudaConvGrad (xgpr_cuda_rfgen_cpp_ext.cpp:81-92)."""
    s = _ftype(inputArr, "inputArr")
    x = _dev(inputArr, "inputArr", None, 3)
    o = _dev(outputArr, "outputArr", torch.float64, 2)
    g = _dev(gradArr, "gradArr", torch.float64, 3)
    r = _radem3(radem)
    c = _dev(chiArr, "chiArr", inputArr.dtype, 1)
    host, dev = _seqlens(seqlengths, inputArr.device)
    ws, wp, wn = _conv_ws(radem, int(convWidth) * inputArr.shape[2], inputArr)
    return _lib.check(getattr(_LIB, f"xgpr_conv_grad_{s}")(
        x, o, g, r, c, C.c_void_p(host.ctypes.data), C.c_void_p(dev.data_ptr()), inputArr.shape[0],
        inputArr.shape[1], inputArr.shape[2], outputArr.shape[0], outputArr.shape[1], gradArr.shape[0],
        gradArr.shape[1], chiArr.shape[0], radem.shape[2], host.shape[0], float(sigma), int(convWidth),
        int(scalingType), wp, wn, _stream()))


@_array_args("inputArr", "outputArr", "radem", "chiArr")
def hipConv1dMaxpool(inputArr, outputArr, radem, chiArr, seqlengths, convWidth):
    """cudaConv1dMaxpool (xgpr_cuda_rfgen_cpp_ext.cpp:61-69); float32 output."""
    s = _ftype(inputArr, "inputArr")
    x = _dev(inputArr, "inputArr", None, 3)
    o = _dev(outputArr, "outputArr", torch.float32, 2)
    r = _radem3(radem)
    c = _dev(chiArr, "chiArr", inputArr.dtype, 1)
    host, dev = _seqlens(seqlengths, inputArr.device)
    ws, wp, wn = _conv_ws(radem, int(convWidth) * inputArr.shape[2], inputArr)
    return _lib.check(getattr(_LIB, f"xgpr_conv1d_maxpool_{s}")(
        x, o, r, c, C.c_void_p(host.ctypes.data), C.c_void_p(dev.data_ptr()), inputArr.shape[0],
        inputArr.shape[1], inputArr.shape[2], outputArr.shape[0], outputArr.shape[1], chiArr.shape[0],
        radem.shape[2], host.shape[0], int(convWidth), wp, wn, _stream()))


@_array_args("inputArr", "radem", "chiArr", "vec", "outVec", "workspace")
def hipZtZMatvec(inputArr, radem, chiArr, vec, outVec, fitIntercept, workspace=None, masksPacked=False):
    """Fused ``Z.T @ (Z @ vec)`` over one shard of (sigma-scaled, float32) rows: the chunk
    body of the reference's CG matvec (fitting_toolkit/cg_tools.py:189-191) with
    ``kernel.transform_x`` fused in.  ``outVec`` [num_rffs] f64 is overwritten.  ``masksPacked``:
    ``workspace`` still holds the sign masks an earlier call packed from the same ``radem``."""
    x = _dev(inputArr, "inputArr", torch.float32, 2)
    r = _radem3(radem)
    c = _dev(chiArr, "chiArr", torch.float32, 1)
    v = _dev(vec, "vec", torch.float64, 1)
    o = _dev(outVec, "outVec", torch.float64, 1)
    if vec.shape[0] != outVec.shape[0]:
        raise TypeError("vec / outVec: shapes differ")
    need = _LIB.xgpr_ztz_matvec_workspace_bytes(outVec.shape[0], radem.shape[2])
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=inputArr.device)
    if masksPacked and workspace is not None:
        r = C.c_void_p(0)
    return _lib.check(_LIB.xgpr_ztz_matvec_f32(
        x, r, c, v, o, inputArr.shape[0], inputArr.shape[1], outVec.shape[0], chiArr.shape[0],
        radem.shape[2], int(bool(fitIntercept)), C.c_void_p(workspace.data_ptr()),
        C.c_size_t(workspace.numel()), _stream()))


@_array_args("inputArr", "radem", "chiArr", "yvec", "outVec", "workspace")
def hipZtY(inputArr, radem, chiArr, yvec, outVec, fitIntercept, workspace=None):
    """Fused ``Z.T @ y`` over one shard (scoring_toolkit/exact_nmll_calcs.py:35-37)."""
    x = _dev(inputArr, "inputArr", torch.float32, 2)
    r = _radem3(radem)
    c = _dev(chiArr, "chiArr", torch.float32, 1)
    y = _dev(yvec, "yvec", torch.float64, 1)
    o = _dev(outVec, "outVec", torch.float64, 1)
    if yvec.shape[0] != inputArr.shape[0]:
        raise TypeError("yvec: one value per datapoint expected")
    need = _LIB.xgpr_ztz_matvec_workspace_bytes(outVec.shape[0], radem.shape[2])
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=inputArr.device)
    return _lib.check(_LIB.xgpr_zty_f32(
        x, r, c, y, o, inputArr.shape[0], inputArr.shape[1], outVec.shape[0], chiArr.shape[0],
        radem.shape[2], int(bool(fitIntercept)), C.c_void_p(workspace.data_ptr()),
        C.c_size_t(workspace.numel()), _stream()))


def _err_ptr(t):
    if t is None:
        return 0
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float64 or t.numel() < 1 or not (t.is_cuda or t.is_pinned()):
        raise TypeError("err_out: expected a float64 tensor on the device or in pinned host memory")
    return t.data_ptr()


def hipCGStep1(w, p, x, r, r_next, z, scal, lam2, init_norm, stop_tol=0.0, err_out=None):
    """cg_tools.py:256-265 for one right-hand side (see include/xgpr_hip.h).  ``stop_tol`` > 0: the
    convergence test is applied on the device too (iterations queued ahead of the host's check);
    scal then has 8 + max_iterations entries.  ``err_out``: a one-element float64 tensor that also receives
    the error -- on the device, or in pinned host memory (written by the kernel itself, no copy command)."""
    for name, t in (("w", w), ("p", p), ("x", x), ("r", r), ("r_next", r_next), ("z", z)):
        _dev(t, name, torch.float64, 1)
    _dev(scal, "scal", torch.float64, 1)
    return _lib.check(_LIB.xgpr_cg_step1_f64(
        C.c_void_p(w.data_ptr()), C.c_void_p(p.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(r.data_ptr()),
        C.c_void_p(r_next.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(scal.data_ptr()), float(lam2),
        float(init_norm), w.shape[0], float(stop_tol), C.c_void_p(_err_ptr(err_out)), _stream()))


def hipCGStep2(r_next, z_next, p, p_next, scal, stop_tol=0.0):
    """cg_tools.py:271-274 for one right-hand side."""
    for name, t in (("r_next", r_next), ("z_next", z_next), ("p", p), ("p_next", p_next)):
        _dev(t, name, torch.float64, 1)
    return _lib.check(_LIB.xgpr_cg_step2_f64(
        C.c_void_p(r_next.data_ptr()), C.c_void_p(z_next.data_ptr()), C.c_void_p(p.data_ptr()),
        C.c_void_p(p_next.data_ptr()), C.c_void_p(scal.data_ptr()), r_next.shape[0], float(stop_tol), _stream()))


def _blk(t, name, k=None):
    _dev(t, name, torch.float64, 2)
    if k is not None and t.shape[1] != k:
        raise TypeError(f"{name}: expected [M, {k}]")
    return C.c_void_p(t.data_ptr())


def hipSoftmaxResidual(pred, labels):
    """nonlinear_cg_toolkit.py:243-262 on the device in one launch: ``pred`` [n, classes] float64 becomes
    softmax_2.71828(pred) - onehot(labels) in place; returns the loss contribution -sum log(max(p[label], 1e-16))
    as a one-element device tensor."""
    _dev(pred, "pred", torch.float64, 2)
    _dev(labels, "labels", torch.int64, 1)
    n, ncls = pred.shape
    if labels.shape[0] != n:
        raise TypeError("labels: expected [n]")
    parts = torch.empty((n + 255) // 256, dtype=torch.float64, device=pred.device)
    _lib.check(_LIB.xgpr_softmax_residual_f64(C.c_void_p(pred.data_ptr()), C.c_void_p(labels.data_ptr()), n, ncls,
                                              C.c_void_p(parts.data_ptr()), _stream()))
    return parts.sum().reshape(1)


CG_BLOCK_MAX_K = 32


def cg_block_workspace_bytes(m, k):
    return int(_LIB.xgpr_cg_block_workspace_bytes(m, k))


def hipCGStep1Block(w, p, x, r, r_next, z, rz, alpha_out, err_out, init_norm, lam2, workspace):
    """cg_tools.py:256-265 for a block of k <= 32 right-hand sides, all [M, k] row-major (see include/xgpr_hip.h).
    ``rz``, ``alpha_out``, ``init_norm``: float64 [k] on the device; ``err_out``: float64 [k], device or pinned host."""
    m, k = w.shape
    ptrs = [_blk(t, n, k) for t, n in ((w, "w"), (p, "p"), (x, "x"), (r, "r"), (r_next, "r_next"), (z, "z"))]
    for t, n in ((rz, "rz"), (alpha_out, "alpha_out"), (init_norm, "init_norm")):
        _dev(t, n, torch.float64, 1)
    if err_out.dtype != torch.float64 or err_out.numel() != k or not err_out.is_contiguous():
        raise TypeError("err_out: expected float64 [k], contiguous")
    return _lib.check(_LIB.xgpr_cg_step1_block_f64(
        *ptrs, C.c_void_p(rz.data_ptr()), C.c_void_p(alpha_out.data_ptr()), C.c_void_p(err_out.data_ptr()),
        C.c_void_p(init_norm.data_ptr()), float(lam2), m, k, C.c_void_p(workspace.data_ptr()),
        C.c_size_t(workspace.numel()), _stream()))


def hipCGStep2Block(r_next, z_next, p, p_next, rz, beta_out, workspace):
    """cg_tools.py:271-274 for a block of right-hand sides."""
    m, k = r_next.shape
    ptrs = [_blk(t, n, k) for t, n in ((r_next, "r_next"), (z_next, "z_next"), (p, "p"), (p_next, "p_next"))]
    _dev(rz, "rz", torch.float64, 1)
    _dev(beta_out, "beta_out", torch.float64, 1)
    return _lib.check(_LIB.xgpr_cg_step2_block_f64(*ptrs, C.c_void_p(rz.data_ptr()), C.c_void_p(beta_out.data_ptr()),
                                                   m, k, C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel()),
                                                   _stream()))


PRECOND_UTR_BLOCK_MAX_K = 32


def precond_utr_block_workspace_bytes(m, rank, k):
    return int(_LIB.xgpr_precond_utr_block_workspace_bytes(m, rank, k))


def hipPrecondUtRBlock(u_mat, rmat, tout, workspace=None):
    """``tout[rank, k] = u_mat.T @ rmat`` for a block of k <= 32 right-hand sides (the first product of
    rand_nys_preconditioners.py:66-72; the library's skinny GEMM takes 15x as long at rank 512, k = 26)."""
    _dev(u_mat, "u_mat", torch.float64, 2)
    _dev(rmat, "rmat", torch.float64, 2)
    _dev(tout, "tout", torch.float64, 2)
    m, rank = u_mat.shape
    k = rmat.shape[1]
    if rmat.shape[0] != m or tuple(tout.shape) != (rank, k):
        raise TypeError("hipPrecondUtRBlock: expected rmat [M, k] and tout [rank, k]")
    if workspace is None:
        workspace = torch.empty(precond_utr_block_workspace_bytes(m, rank, k), dtype=torch.uint8, device=u_mat.device)
    return _lib.check(_LIB.xgpr_precond_utr_block_f64(
        C.c_void_p(u_mat.data_ptr()), C.c_void_p(rmat.data_ptr()), C.c_void_p(tout.data_ptr()), m, rank, k,
        C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel()), _stream()))


def precond_apply_block_workspace_bytes(m, rank, k):
    return int(_LIB.xgpr_precond_apply_block_workspace_bytes(m, rank, k))


def hipPrecondApplyBlock(u_mat, inv_eig, prefactor, rmat, zmat, workspace=None):
    """RandNysPreconditioner.batch_matvec for a block of k <= 32 right-hand sides (rand_nys_preconditioners.py:66-72):
    ``zmat <- rmat + U ((inv_eig * prefactor - 1) .* (U^T rmat))``, both products on the float64 matrix cores."""
    _dev(u_mat, "u_mat", torch.float64, 2)
    _dev(inv_eig, "inv_eig", torch.float64, 1)
    _dev(rmat, "rmat", torch.float64, 2)
    _dev(zmat, "zmat", torch.float64, 2)
    m, rank = u_mat.shape
    k = rmat.shape[1]
    if rmat.shape[0] != m or tuple(zmat.shape) != (m, k) or inv_eig.shape[0] != rank or zmat.data_ptr() == rmat.data_ptr():
        raise TypeError("hipPrecondApplyBlock: expected rmat, zmat [M, k] (distinct) and inv_eig [rank]")
    if workspace is None:
        workspace = torch.empty(precond_apply_block_workspace_bytes(m, rank, k), dtype=torch.uint8, device=u_mat.device)
    return _lib.check(_LIB.xgpr_precond_apply_block_f64(
        C.c_void_p(u_mat.data_ptr()), C.c_void_p(inv_eig.data_ptr()), float(prefactor), C.c_void_p(rmat.data_ptr()),
        C.c_void_p(zmat.data_ptr()), m, rank, k, C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel()), _stream()))


def hipPrecondApply(u_mat, inv_eig, prefactor, rvec, zvec, workspace=None):
    """RandNysPreconditioner.batch_matvec for one right-hand side
    (preconditioners/rand_nys_preconditioners.py:66-72): zvec <- P^-1 rvec."""
    _dev(u_mat, "u_mat", torch.float64, 2)
    _dev(inv_eig, "inv_eig", torch.float64, 1)
    _dev(rvec, "rvec", torch.float64, 1)
    _dev(zvec, "zvec", torch.float64, 1)
    if u_mat.shape[0] != rvec.shape[0] or u_mat.shape[1] != inv_eig.shape[0] or zvec.shape[0] != rvec.shape[0]:
        raise TypeError("hipPrecondApply: shapes do not match")
    need = _LIB.xgpr_precond_apply_workspace_bytes(u_mat.shape[1])
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=u_mat.device)
    return _lib.check(_LIB.xgpr_precond_apply_f64(
        C.c_void_p(u_mat.data_ptr()), C.c_void_p(inv_eig.data_ptr()), float(prefactor),
        C.c_void_p(rvec.data_ptr()), C.c_void_p(zvec.data_ptr()), u_mat.shape[0], u_mat.shape[1],
        C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel()), _stream()))


def precond_workspace_bytes(rank):
    return int(_LIB.xgpr_precond_apply_workspace_bytes(rank))


def hipRBFFeatureCache(inputArr, cacheArr, radem, chiArr):
    """float32 (cos, sin) pairs before scaling, cacheArr [N, num_rffs] float32 (see include/xgpr_hip.h)."""
    x = _dev(inputArr, "inputArr", torch.float32, 2)
    zc = _dev(cacheArr, "cacheArr", torch.float32, 2)
    r = _radem3(radem)
    c = _dev(chiArr, "chiArr", torch.float32, 1)
    if cacheArr.shape[0] != inputArr.shape[0]:
        raise RuntimeError("no datapoints")
    ws, wp, wn = _sorf_ws(radem, inputArr.shape[1], inputArr)
    return _lib.check(_LIB.xgpr_rbf_feature_cache_f32(
        x, zc, r, c, inputArr.shape[0], inputArr.shape[1], cacheArr.shape[1], chiArr.shape[0], radem.shape[2],
        wp, wn, _stream()))


def hipZCacheMatvecScaled(cacheArr, vec, outVec, scale, workspace):
    """``Z.T @ (Z @ vec)`` with Z = scale * cacheArr (any kernel's float32 feature rows)."""
    zc = _dev(cacheArr, "cacheArr", torch.float32, 2)
    v = _dev(vec, "vec", torch.float64, 1)
    o = _dev(outVec, "outVec", torch.float64, 1)
    if vec.shape[0] != cacheArr.shape[1] or outVec.shape[0] != cacheArr.shape[1]:
        raise TypeError("vec / outVec: expected num_rffs entries")
    return _lib.check(_LIB.xgpr_zcache_matvec_scaled_f32(
        zc, v, o, cacheArr.shape[0], cacheArr.shape[1], float(scale),
        C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel()), _stream()))


def hipZCacheMatvec(cacheArr, vec, outVec, fitIntercept, workspace):
    """``Z.T @ (Z @ vec)`` streamed from the resident feature cache (cg_tools.py:189-191)."""
    zc = _dev(cacheArr, "cacheArr", torch.float32, 2)
    v = _dev(vec, "vec", torch.float64, 1)
    o = _dev(outVec, "outVec", torch.float64, 1)
    if vec.shape[0] != cacheArr.shape[1] or outVec.shape[0] != cacheArr.shape[1]:
        raise TypeError("vec / outVec: expected num_rffs entries")
    return _lib.check(_LIB.xgpr_zcache_matvec_f32(
        zc, v, o, cacheArr.shape[0], cacheArr.shape[1], int(bool(fitIntercept)),
        C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel()), _stream()))


def hipZCacheBlockMatvec(cacheArr, vecs, outVecs, fitIntercept, workspace, scale=0.0, accumulate=False):
    """``outVecs (+)= Z.T @ (Z @ vecs)`` for vecs [num_rffs, k <= 32] on the float64 matrix cores, Z streamed
    from the resident feature cache (cg_tools.py:41-44 with a block of right-hand sides).  scale = 0 selects
    the RBF-family scale; a positive scale is for caches that already hold complete feature rows / scale."""
    zc = _dev(cacheArr, "cacheArr", torch.float32, 2)
    v = _dev(vecs, "vecs", torch.float64, 2)
    o = _dev(outVecs, "outVecs", torch.float64, 2)
    if vecs.shape[0] != cacheArr.shape[1] or tuple(outVecs.shape) != tuple(vecs.shape):
        raise TypeError("vecs / outVecs: expected [num_rffs, k]")
    return _lib.check(_LIB.xgpr_zcache_block_matvec_f32(
        zc, v, o, cacheArr.shape[0], cacheArr.shape[1], vecs.shape[1], int(bool(fitIntercept)), float(scale),
        int(bool(accumulate)), C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel()), _stream()))


def hipZCacheBlockProject(cacheArr, vecs, outArr, fitIntercept, scale=0.0, workspace=None):
    """``outArr[n, k] = Z @ vecs`` (the classifier's ``xd @ wvec``, nonlinear_cg_toolkit.py:251) on the
    float64 matrix cores over the float32 feature rows.  ``workspace`` (zcache_block_project_workspace_bytes; allocated
    here when a short launch would use one and none is passed) lets launches of fewer than 65536 rows split the
    contraction over the features across workgroups."""
    zc = _dev(cacheArr, "cacheArr", torch.float32, 2)
    v = _dev(vecs, "vecs", torch.float64, 2)
    o = _dev(outArr, "outArr", torch.float64, 2)
    if vecs.shape[0] != cacheArr.shape[1] or tuple(outArr.shape) != (cacheArr.shape[0], vecs.shape[1]):
        raise TypeError("vecs: expected [num_rffs, k]; outArr: expected [n, k]")
    if workspace is None:
        need = zcache_block_project_workspace_bytes(cacheArr.shape[0], cacheArr.shape[1], vecs.shape[1])
        if need:
            workspace = torch.empty(need, dtype=torch.uint8, device=cacheArr.device)
    wp, wn = (C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel())) if workspace is not None else (None, C.c_size_t(0))
    return _lib.check(_LIB.xgpr_zcache_block_project_f32(
        zc, v, o, cacheArr.shape[0], cacheArr.shape[1], vecs.shape[1], int(bool(fitIntercept)), float(scale),
        wp, wn, _stream()))


def hipZCacheBlockBackproject(cacheArr, resid, outVecs, fitIntercept, workspace, scale=0.0, accumulate=False):
    """``outVecs[num_rffs, k] (+)= Z.T @ resid`` (the classifier's gradient sums,
    nonlinear_cg_toolkit.py:264-269)."""
    zc = _dev(cacheArr, "cacheArr", torch.float32, 2)
    r = _dev(resid, "resid", torch.float64, 2)
    o = _dev(outVecs, "outVecs", torch.float64, 2)
    if resid.shape[0] != cacheArr.shape[0] or tuple(outVecs.shape) != (cacheArr.shape[1], resid.shape[1]):
        raise TypeError("resid: expected [n, k]; outVecs: expected [num_rffs, k]")
    return _lib.check(_LIB.xgpr_zcache_block_backproject_f32(
        zc, r, o, cacheArr.shape[0], cacheArr.shape[1], resid.shape[1], int(bool(fitIntercept)), float(scale),
        int(bool(accumulate)), C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel()), _stream()))


def zcache_block_project_workspace_bytes(ndatapoints, num_rffs, k):
    return int(_LIB.xgpr_zcache_block_project_workspace_bytes(ndatapoints, num_rffs, k))


def zcache_block_workspace_bytes(ndatapoints, num_rffs, k):
    return int(_LIB.xgpr_zcache_block_workspace_bytes(ndatapoints, num_rffs, k))


def ztz_workspace_bytes(num_rffs, radem_shape2):
    return int(_LIB.xgpr_ztz_matvec_workspace_bytes(num_rffs, radem_shape2))


def ztz_matvec_plan(d, num_freqs):
    """1: single pass on the three-wave kernel, 2: single pass on the two-wave kernel, 3: two feature passes, 0: unsupported
    (xgpr_ztz_matvec_plan; no device work)."""
    return int(_LIB.xgpr_ztz_matvec_plan(int(d), int(num_freqs)))


def selftest_lane_xor(device="cuda"):
    """Runs the cross-lane butterfly self test; returns an int32 [6, 16, 64] CPU array."""
    out = torch.zeros(6 * 16 * 64, dtype=torch.int32, device=device)
    _lib.check(_LIB.xgpr_selftest_lane_xor(C.c_void_p(out.data_ptr()), _stream()))
    return out.cpu().numpy().reshape(6, 16, 64)


# the reference's names, so that its kernel classes / tests can bind to this module unchanged
cudaFastHadamardTransform2D = hipFastHadamardTransform2D
cudaSRHT = hipSRHT
cudaRBFFeatureGen = hipRBFFeatureGen
cudaRBFGrad = hipRBFGrad
cudaMiniARDGrad = hipMiniARDGrad
cudaConv1dMaxpool = hipConv1dMaxpool
cudaConv1dFGen = hipConv1dFGen
cudaConvGrad = hipConvGrad
