"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy/ctypes front-end of the CPU restatement of the reference hot path:

* ``Oracle``   -- ctypes binding of ``oracle/liboracle.so`` (xgpr_oracle.c, the
  plain-C restatement of the reference's native ops), validating and raising
  ``RuntimeError`` exactly where the reference throws.
* ``RefCore``  -- ctypes binding of ``oracle/_ref/libxgpr_ref.so`` (the
  reference's own compiled arithmetic core; exists only where oracle/Makefile
  could see /root/reference).  Used to pin ``Oracle``.
* numpy restatements of the reference's *host-side* steps of the path: kernel
  parameter draws, ``transform_x``, the preconditioned CG solver, the
  randomized-Nystrom (SRHT) preconditioner.  Each cites the reference
  file:line (relative to /root/reference/src/xGPR/).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product (``xgpr_amd``) never does.

Parity status: pinned -- see tests/test_oracle_vs_ref.py (against the compiled
reference core) and tests/test_oracle_golden.py (against tests/golden/*.npz,
produced by the reference itself with tests/golden/make_golden.py).
"""
import ctypes as C
import os
import subprocess
from math import ceil

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

_ERRORS = {
    -1: "no datapoints",
    -2: "last dim of output must be even number",
    -3: "incorrect number of rffs and or freqs.",
    -4: "Wrong array sizes.",
    -5: "wrong array sizes",
    -6: "invalid conv_width",
    -7: "All sequence lengths must be >= conv width and < array size.",
    -8: "incorrect array dims passed",
    -9: "last dim not power of 2 > 1",
}


def build(ref=True):
    """Compile liboracle.so (and _ref when the reference tree is present)."""
    subprocess.run(["make", "-s", "-C", _HERE, "all" if ref else "oracle"],
                   check=True)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _check(a, dtype, ndim, name):
    if not isinstance(a, np.ndarray) or a.dtype != dtype or a.ndim != ndim \
            or not a.flags["C_CONTIGUOUS"]:
        raise TypeError(f"{name}: expected C-contiguous {np.dtype(dtype).name} "
                        f"array with {ndim} dims")


def _suffix(x):
    if x.dtype == np.float32:
        return "f32"
    if x.dtype == np.float64:
        return "f64"
    raise TypeError("input must be float32 or float64")


class Oracle:
    """ctypes binding of the C restatement; the operator names follow the
    reference's extension module (cpu_rf_gen/xgpr_cpu_rfgen_cpp_ext.cpp:24-146)."""

    def __init__(self):
        # XGPR_ORACLE_LIB: an alternative build of the same source (the sanitizer job of
        # tests/test_oracle_sanitizer.py loads oracle/_asan/liboracle_asan.so through it)
        path = os.environ.get("XGPR_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        self.lib = C.CDLL(path)
        self.lib.orc_num_threads.restype = C.c_int

    def num_threads(self):
        return int(self.lib.orc_num_threads())

    def _call(self, name, *args):
        fn = getattr(self.lib, name)
        fn.restype = C.c_int
        rc = fn(*args)
        if rc != 0:
            raise RuntimeError(_ERRORS.get(rc, f"oracle error {rc}"))
        return 0

    # cpuFastHadamardTransform (3-D) / cpuFastHadamardTransform2D
    def cpuFastHadamardTransform(self, inputArr):
        _check(inputArr, inputArr.dtype, 3, "inputArr")
        s = _suffix(inputArr)
        return self._call(f"orc_fht_{s}", _ptr(inputArr), C.c_long(inputArr.shape[0]),
                          C.c_long(inputArr.shape[1]), C.c_long(inputArr.shape[2]))

    def cpuFastHadamardTransform2D(self, inputArr):
        _check(inputArr, inputArr.dtype, 2, "inputArr")
        s = _suffix(inputArr)
        return self._call(f"orc_fht_{s}", _ptr(inputArr), C.c_long(inputArr.shape[0]),
                          C.c_long(1), C.c_long(inputArr.shape[1]))

    def cpuSRHT(self, inputArr, radem):
        _check(inputArr, inputArr.dtype, 2, "inputArr")
        _check(radem, np.int8, 1, "radem")
        s = _suffix(inputArr)
        return self._call(f"orc_srht_{s}", _ptr(inputArr), _ptr(radem),
                          C.c_long(inputArr.shape[0]), C.c_long(inputArr.shape[1]),
                          C.c_long(radem.shape[0]))

    def cpuRBFFeatureGen(self, inputArr, outputArr, radem, chiArr, fitIntercept):
        s = _suffix(inputArr)
        _check(inputArr, inputArr.dtype, 2, "inputArr")
        _check(outputArr, np.float64, 2, "outputArr")
        _check(radem, np.int8, 3, "radem")
        _check(chiArr, inputArr.dtype, 1, "chiArr")
        if radem.shape[0] != 3 or radem.shape[1] != 1:
            raise TypeError("radem must have shape (3,1,R)")
        return self._call(f"orc_rbf_feature_gen_{s}", _ptr(inputArr), _ptr(outputArr),
                          _ptr(radem), _ptr(chiArr), C.c_long(inputArr.shape[0]),
                          C.c_int(inputArr.shape[1]), C.c_long(outputArr.shape[0]),
                          C.c_long(outputArr.shape[1]), C.c_long(chiArr.shape[0]),
                          C.c_long(radem.shape[2]), C.c_int(bool(fitIntercept)))

    def cpuRBFGrad(self, inputArr, outputArr, gradArr, radem, chiArr, sigma, fitIntercept):
        s = _suffix(inputArr)
        _check(inputArr, inputArr.dtype, 2, "inputArr")
        _check(outputArr, np.float64, 2, "outputArr")
        _check(gradArr, np.float64, 3, "gradArr")
        _check(radem, np.int8, 3, "radem")
        _check(chiArr, inputArr.dtype, 1, "chiArr")
        return self._call(f"orc_rbf_grad_{s}", _ptr(inputArr), _ptr(outputArr), _ptr(gradArr),
                          _ptr(radem), _ptr(chiArr), C.c_long(inputArr.shape[0]),
                          C.c_int(inputArr.shape[1]), C.c_long(outputArr.shape[0]),
                          C.c_long(outputArr.shape[1]), C.c_long(gradArr.shape[0]),
                          C.c_long(gradArr.shape[1]), C.c_long(chiArr.shape[0]),
                          C.c_long(radem.shape[2]), C.c_double(float(sigma)),
                          C.c_int(bool(fitIntercept)))

    def cpuMiniARDGrad(self, inputArr, outputArr, precompWeights, sigmaMap, sigmaVals, gradArr, fitIntercept):
        s = _suffix(inputArr)
        _check(inputArr, inputArr.dtype, 2, "inputArr")
        _check(outputArr, np.float64, 2, "outputArr")
        _check(precompWeights, inputArr.dtype, 2, "precompWeights")
        _check(sigmaMap, np.int32, 1, "sigmaMap")
        _check(sigmaVals, np.float64, 1, "sigmaVals")
        _check(gradArr, np.float64, 3, "gradArr")
        return self._call(f"orc_mini_ard_grad_{s}", _ptr(inputArr), _ptr(outputArr), _ptr(precompWeights),
                          _ptr(sigmaMap), _ptr(sigmaVals), _ptr(gradArr), C.c_long(inputArr.shape[0]),
                          C.c_long(inputArr.shape[1]), C.c_long(outputArr.shape[0]),
                          C.c_long(outputArr.shape[1]), C.c_long(precompWeights.shape[0]),
                          C.c_long(precompWeights.shape[1]), C.c_long(sigmaMap.shape[0]),
                          C.c_long(sigmaVals.shape[0]), C.c_long(gradArr.shape[0]),
                          C.c_long(gradArr.shape[1]), C.c_long(gradArr.shape[2]),
                          C.c_int(bool(fitIntercept)))

    def cpuConv1dFGen(self, inputArr, outputArr, radem, chiArr, seqlengths, convWidth, scalingType):
        s = _suffix(inputArr)
        _check(inputArr, inputArr.dtype, 3, "inputArr")
        _check(outputArr, np.float64, 2, "outputArr")
        _check(radem, np.int8, 3, "radem")
        _check(chiArr, inputArr.dtype, 1, "chiArr")
        _check(seqlengths, np.int32, 1, "seqlengths")
        return self._call(f"orc_conv1d_fgen_{s}", _ptr(inputArr), _ptr(outputArr), _ptr(radem),
                          _ptr(chiArr), _ptr(seqlengths), C.c_long(inputArr.shape[0]),
                          C.c_int(inputArr.shape[1]), C.c_int(inputArr.shape[2]),
                          C.c_long(outputArr.shape[0]), C.c_long(outputArr.shape[1]),
                          C.c_long(chiArr.shape[0]), C.c_long(radem.shape[2]),
                          C.c_long(seqlengths.shape[0]), C.c_int(int(convWidth)),
                          C.c_int(int(scalingType)))

    def cpuConvGrad(self, inputArr, outputArr, radem, chiArr, seqlengths, gradArr, sigma,
                    convWidth, scalingType):
        s = _suffix(inputArr)
        _check(inputArr, inputArr.dtype, 3, "inputArr")
        _check(outputArr, np.float64, 2, "outputArr")
        _check(gradArr, np.float64, 3, "gradArr")
        _check(radem, np.int8, 3, "radem")
        _check(chiArr, inputArr.dtype, 1, "chiArr")
        _check(seqlengths, np.int32, 1, "seqlengths")
        return self._call(f"orc_conv_grad_{s}", _ptr(inputArr), _ptr(outputArr), _ptr(gradArr),
                          _ptr(radem), _ptr(chiArr), _ptr(seqlengths),
                          C.c_long(inputArr.shape[0]), C.c_int(inputArr.shape[1]),
                          C.c_int(inputArr.shape[2]), C.c_long(outputArr.shape[0]),
                          C.c_long(outputArr.shape[1]), C.c_long(gradArr.shape[0]),
                          C.c_long(gradArr.shape[1]), C.c_long(chiArr.shape[0]),
                          C.c_long(radem.shape[2]), C.c_long(seqlengths.shape[0]),
                          C.c_double(float(sigma)), C.c_int(int(convWidth)),
                          C.c_int(int(scalingType)))

    def cpuConv1dMaxpool(self, inputArr, outputArr, radem, chiArr, seqlengths, convWidth):
        s = _suffix(inputArr)
        _check(inputArr, inputArr.dtype, 3, "inputArr")
        _check(outputArr, np.float32, 2, "outputArr")
        _check(radem, np.int8, 3, "radem")
        _check(chiArr, inputArr.dtype, 1, "chiArr")
        _check(seqlengths, np.int32, 1, "seqlengths")
        return self._call(f"orc_conv1d_maxpool_{s}", _ptr(inputArr), _ptr(outputArr), _ptr(radem),
                          _ptr(chiArr), _ptr(seqlengths), C.c_long(inputArr.shape[0]),
                          C.c_int(inputArr.shape[1]), C.c_int(inputArr.shape[2]),
                          C.c_long(outputArr.shape[0]), C.c_long(outputArr.shape[1]),
                          C.c_long(chiArr.shape[0]), C.c_long(radem.shape[2]),
                          C.c_long(seqlengths.shape[0]), C.c_int(int(convWidth)))

    def ztz_matvec(self, z, v, w):
        """w += Z^T (Z v) (cg_tools.py:189-191 for k = 1), OpenMP; bench baseline only."""
        fn = self.lib.orc_ztz_matvec_f64
        fn.restype = None
        fn(_ptr(z), _ptr(v), _ptr(w), C.c_long(z.shape[0]), C.c_long(z.shape[1]))


class RefCore:
    """ctypes binding of the reference's compiled core (authoring container only).
    No validation: callers pass well-formed arrays."""

    def __init__(self):
        path = os.path.join(_HERE, "_ref", "libxgpr_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libxgpr_ref.so"))

    def _v(self, name, *args):
        fn = getattr(self.lib, name)
        fn.restype = None
        fn(*args)
        return 0

    def cpuFastHadamardTransform(self, x):
        return self._v(f"ref_fht_rows_{_suffix(x)}", _ptr(x), C.c_int(x.shape[0]),
                       C.c_int(x.shape[1]), C.c_int(x.shape[2]))

    def cpuFastHadamardTransform2D(self, x):
        return self._v(f"ref_fht_rows_{_suffix(x)}", _ptr(x), C.c_int(x.shape[0]),
                       C.c_int(1), C.c_int(x.shape[1]))

    def cpuSRHT(self, x, radem):
        return self._v(f"ref_srht_{_suffix(x)}", _ptr(x), _ptr(radem), C.c_int(x.shape[0]),
                       C.c_int(x.shape[1]))

    def cpuRBFFeatureGen(self, x, out, radem, chi, fitIntercept):
        return self._v(f"ref_rbf_feature_gen_{_suffix(x)}", _ptr(x), _ptr(out), _ptr(radem),
                       _ptr(chi), C.c_long(x.shape[0]), C.c_int(x.shape[1]),
                       C.c_long(chi.shape[0]), C.c_long(radem.shape[2]),
                       C.c_int(bool(fitIntercept)))

    def cpuRBFGrad(self, x, out, grad, radem, chi, sigma, fitIntercept):
        return self._v(f"ref_rbf_grad_{_suffix(x)}", _ptr(x), _ptr(out), _ptr(grad), _ptr(radem),
                       _ptr(chi), C.c_long(x.shape[0]), C.c_int(x.shape[1]),
                       C.c_long(chi.shape[0]), C.c_long(radem.shape[2]),
                       C.c_double(float(sigma)), C.c_int(bool(fitIntercept)))

    def cpuConv1dFGen(self, x, out, radem, chi, seqlen, convWidth, scalingType):
        return self._v(f"ref_conv1d_fgen_{_suffix(x)}", _ptr(x), _ptr(out), _ptr(radem),
                       _ptr(chi), _ptr(seqlen), C.c_long(x.shape[0]), C.c_int(x.shape[1]),
                       C.c_int(x.shape[2]), C.c_long(chi.shape[0]), C.c_long(radem.shape[2]),
                       C.c_int(int(convWidth)), C.c_int(int(scalingType)))

    def cpuConvGrad(self, x, out, radem, chi, seqlen, grad, sigma, convWidth, scalingType):
        return self._v(f"ref_conv_grad_{_suffix(x)}", _ptr(x), _ptr(out), _ptr(grad), _ptr(radem),
                       _ptr(chi), _ptr(seqlen), C.c_long(x.shape[0]), C.c_int(x.shape[1]),
                       C.c_int(x.shape[2]), C.c_long(chi.shape[0]), C.c_long(radem.shape[2]),
                       C.c_double(float(sigma)), C.c_int(int(convWidth)),
                       C.c_int(int(scalingType)))

    def cpuConv1dMaxpool(self, x, out, radem, chi, seqlen, convWidth):
        return self._v(f"ref_conv1d_maxpool_{_suffix(x)}", _ptr(x), _ptr(out), _ptr(radem),
                       _ptr(chi), _ptr(seqlen), C.c_long(x.shape[0]), C.c_int(x.shape[1]),
                       C.c_int(x.shape[2]), C.c_long(chi.shape[0]), C.c_int(int(convWidth)))


# --------------------------------------------------------------------------
# Host-side restatements (numpy).  Paths relative to /root/reference/src/xGPR/.
# --------------------------------------------------------------------------

def padded_dims(width):
    """kernels/basic_kernels/sorf_kernel_baseclass.py:71"""
    return 2 ** ceil(np.log2(max(width, 2)))


def draw_sorf_params(num_rffs, xwidth, random_seed=123, double_precision=False,
                     conv=False):
    """radem_diag / chi_arr draws.
    Fixed-vector: kernels/basic_kernels/sorf_kernel_baseclass.py:71-84.
    Convolution (xwidth = conv_width * C): kernels/convolution_kernels/
    conv_kernel_baseclass.py:85-99 (the radem length is always rounded up there,
    and the rng is created before being used for radem only)."""
    from scipy.stats import chi
    num_freqs = num_rffs // 2
    pdims = padded_dims(xwidth)
    radem_array = np.asarray([-1, 1], dtype=np.int8)
    rng = np.random.default_rng(random_seed)
    if conv:
        rlen = ceil(num_freqs / pdims) * pdims
    else:
        nblocks = ceil(num_freqs / pdims) if pdims < num_freqs else 1
        rlen = nblocks * pdims
    radem = rng.choice(radem_array, size=(3, 1, rlen), replace=True)
    chi_arr = chi.rvs(df=pdims, size=num_freqs, random_state=random_seed)
    if not double_precision:
        chi_arr = chi_arr.astype(np.float32)
    return radem, chi_arr


def matern_rescale(chi_arr, nu, random_seed=123):
    """kernels/basic_kernels/matern.py:50-54 (in place on the stored dtype)."""
    rng = np.random.default_rng(random_seed)
    chisamples = np.sqrt(rng.chisquare(2 * nu, size=chi_arr.shape[0]) / (nu * 2))
    chi_arr /= chisamples
    return chi_arr


def cauchy_rescale(chi_arr, random_seed=123):
    """kernels/basic_kernels/cauchy.py:39-41."""
    rng = np.random.default_rng(random_seed)
    dstsamples = np.sqrt(rng.exponential(size=chi_arr.shape[0]))
    chi_arr *= dstsamples
    return chi_arr


def draw_srht_params(compression_size, input_size, random_seed=123):
    """kernels/srht_compressor.py:55-66 -> (radem[Pm] int8, col_sampler[Pm] int64)."""
    pdims = padded_dims(input_size)
    radem_array = np.asarray([-1, 1], dtype=np.int8)
    rng = np.random.default_rng(random_seed)
    radem = rng.choice(radem_array, size=(pdims), replace=True)
    col_sampler = rng.permutation(pdims)
    return radem, col_sampler


class OracleKernel:
    """Restates KernelBaseclass.transform_x (kernels/kernel_baseclass.py:269-299) +
    SORFKernelBaseclass.kernel_specific_transform (sorf_kernel_baseclass.py:104-126)
    / ConvKernelBaseclass.kernel_specific_transform (conv_kernel_baseclass.py:116-147)
    on top of the C oracle."""

    def __init__(self, kind, num_rffs, xdim, hyperparams, random_seed=123,
                 matern_nu=2.5, conv_width=9, averaging="none", fit_intercept=True,
                 ops=None):
        self.ops = ops if ops is not None else Oracle()
        self.kind = kind
        self.num_rffs = num_rffs
        self.num_freqs = num_rffs // 2
        self.hyperparams = np.asarray(hyperparams, dtype=np.float64)
        self.fit_intercept = fit_intercept
        self.conv = len(xdim) == 3
        self.conv_width = conv_width
        self.scaling_type = {"none": 0, "sqrt": 1, "full": 2}[averaging]
        width = conv_width * xdim[2] if self.conv else xdim[-1]
        self.radem_diag, self.chi_arr = draw_sorf_params(num_rffs, width, random_seed,
                                                         conv=self.conv)
        if kind.lower().endswith("matern"):
            matern_rescale(self.chi_arr, matern_nu, random_seed)
        elif kind.lower().endswith("cauchy"):
            cauchy_rescale(self.chi_arr, random_seed)

    def get_lambda(self):
        return self.hyperparams[0]

    def get_num_rffs(self):
        return self.num_rffs

    def transform_x(self, input_x, sequence_length=None):
        xin = input_x.astype(np.float32, copy=True)
        if not xin.flags["C_CONTIGUOUS"]:
            xin = np.ascontiguousarray(xin)
        xin *= self.hyperparams[1]
        out = np.zeros((xin.shape[0], self.num_rffs), np.float64)
        if self.conv:
            slen = sequence_length.astype(np.int32, copy=False)
            self.ops.cpuConv1dFGen(xin, out, self.radem_diag, self.chi_arr, slen,
                                   self.conv_width, self.scaling_type)
        else:
            self.ops.cpuRBFFeatureGen(xin, out, self.radem_diag, self.chi_arr,
                                      self.fit_intercept)
        if self.fit_intercept:
            out[:, 0] = 1.
        return out

    def gradient_x(self, input_x, sequence_length=None):
        """kernels/kernel_baseclass.py:328-361 with sorf_kernel_baseclass.py:136-162 /
        conv_kernel_baseclass.py:157-190: unscaled float32 input, sigma passed to the operator."""
        xin = np.ascontiguousarray(input_x.astype(np.float32, copy=True))
        out = np.zeros((xin.shape[0], self.num_rffs), np.float64)
        grad = np.zeros((xin.shape[0], self.num_rffs, 1), np.float64)
        if self.conv:
            slen = sequence_length.astype(np.int32, copy=False)
            self.ops.cpuConvGrad(xin, out, self.radem_diag, self.chi_arr, slen, grad,
                                 self.hyperparams[1], self.conv_width, self.scaling_type)
        else:
            self.ops.cpuRBFGrad(xin, out, grad, self.radem_diag, self.chi_arr,
                                self.hyperparams[1], self.fit_intercept)
        if self.fit_intercept:
            out[:, 0] = 1.
            grad[:, 0, :] = 0.
        return out, grad


class OracleTwoLayerKernel:
    """kernels/convolution_kernels/l2_conv1d.py:16-222: max-pooled convolution features (cpuConv1dMaxpool),
    scaled by sigma, into an RBF feature map (cpuRBFFeatureGen / cpuRBFGrad)."""

    def __init__(self, num_rffs, xdim, hyperparams, conv_width, init_rffs, random_seed=123, fit_intercept=True, ops=None):
        from scipy.stats import chi as _chi
        self.ops = ops if ops is not None else Oracle()
        self.num_rffs, self.num_freqs = num_rffs, num_rffs // 2
        self.hyperparams = np.asarray(hyperparams, dtype=np.float64)
        self.fit_intercept, self.conv_width, self.init_rffs = fit_intercept, conv_width, init_rffs
        rng = np.random.default_rng(random_seed)
        pd1 = padded_dims(conv_width * xdim[2])
        radem_array = np.asarray([-1, 1], dtype=np.int8)
        self.radem_diag1 = rng.choice(radem_array, size=(3, 1, ceil(init_rffs / pd1) * pd1), replace=True)
        self.chi_arr1 = _chi.rvs(df=pd1, size=init_rffs, random_state=random_seed).astype(np.float32)
        pd2 = padded_dims(init_rffs)
        nblocks = ceil(self.num_freqs / pd2) if pd2 < self.num_freqs else 1
        self.radem_diag2 = rng.choice(radem_array, size=(3, 1, nblocks * pd2), replace=True)
        self.chi_arr2 = _chi.rvs(df=pd2, size=self.num_freqs, random_state=random_seed).astype(np.float32)

    def get_lambda(self):
        return self.hyperparams[0]

    def get_num_rffs(self):
        return self.num_rffs

    def _first_layer(self, input_x, sequence_length):
        xin = np.ascontiguousarray(input_x.astype(np.float32, copy=True))
        feat = np.zeros((xin.shape[0], self.init_rffs), np.float32)
        self.ops.cpuConv1dMaxpool(xin, feat, self.radem_diag1, self.chi_arr1,
                                  sequence_length.astype(np.int32, copy=False), self.conv_width)
        return feat

    def transform_x(self, input_x, sequence_length=None):
        feat = self._first_layer(input_x, sequence_length)
        feat *= self.hyperparams[1]
        out = np.zeros((feat.shape[0], self.num_rffs), np.float64)
        self.ops.cpuRBFFeatureGen(feat, out, self.radem_diag2, self.chi_arr2, self.fit_intercept)
        if self.fit_intercept:
            out[:, 0] = 1.
        return out

    def gradient_x(self, input_x, sequence_length=None):
        feat = self._first_layer(input_x, sequence_length)
        out = np.zeros((feat.shape[0], self.num_rffs), np.float64)
        grad = np.zeros((feat.shape[0], self.num_rffs, 1), np.float64)
        self.ops.cpuRBFGrad(feat, out, grad, self.radem_diag2, self.chi_arr2, self.hyperparams[1], self.fit_intercept)
        if self.fit_intercept:
            out[:, 0] = 1.
            grad[:, 0, :] = 0.
        return out, grad


class OracleMiniARDKernel:
    """kernels/ARD_kernels/mini_ard.py:16-287: one inverse lengthscale per group of input features.
    Features: the input scaled per feature, then the SORF operator (:171-194); gradient: dense
    precomputed weights (three FHT rounds applied to the identity, :196-238) through
    cpuMiniARDGrad (:240-275)."""

    def __init__(self, num_rffs, xdim, split_points, hyperparams=None, random_seed=123,
                 double_precision=False, fit_intercept=True, ops=None):
        self.ops = ops if ops is not None else Oracle()
        self.num_rffs, self.num_freqs = num_rffs, num_rffs // 2
        self.fit_intercept = fit_intercept
        self.double_precision = double_precision
        self.xdim = tuple(xdim)
        self.split_pts = np.sort([0] + list(split_points) + [xdim[1]])
        self.hyperparams = np.ones((self.split_pts.shape[0])) if hyperparams is None \
            else np.asarray(hyperparams, dtype=np.float64)
        self.padded_dims = padded_dims(xdim[-1])
        self.nblocks = ceil(self.num_freqs / self.padded_dims) if self.padded_dims < self.num_freqs else 1
        rng = np.random.default_rng(random_seed)
        self.radem_diag = rng.choice(np.asarray([-1, 1], dtype=np.int8),
                                     size=(3, 1, self.nblocks * self.padded_dims), replace=True)
        from scipy.stats import chi as _chi
        self.chi_arr = _chi.rvs(df=self.padded_dims, size=self.num_freqs, random_state=random_seed)
        if not double_precision:
            self.chi_arr = self.chi_arr.astype(np.float32)
        self.full_ard_weights = np.zeros((xdim[-1]))
        self.ard_position_key = np.zeros((xdim[-1]), dtype=np.int32)
        for i in range(1, self.split_pts.shape[0]):
            self.full_ard_weights[self.split_pts[i - 1]:self.split_pts[i]] = self.hyperparams[i]
            self.ard_position_key[self.split_pts[i - 1]:self.split_pts[i]] = i - 1
        self.precomputed_weights = None

    def get_lambda(self):
        return self.hyperparams[0]

    def get_num_rffs(self):
        return self.num_rffs

    def transform_x(self, input_x, sequence_length=None):
        xtrans = input_x * self.full_ard_weights[None, :]
        xtrans = np.ascontiguousarray(xtrans.astype(np.float64 if self.double_precision else np.float32))
        out = np.zeros((input_x.shape[0], self.num_rffs), np.float64)
        self.ops.cpuRBFFeatureGen(xtrans, out, self.radem_diag, self.chi_arr, self.fit_intercept)
        if self.fit_intercept:
            out[:, 0] = 1.
        return out

    def precompute_weights(self):
        nc = 1.0 / (2.0 ** (np.log2(self.padded_dims) / 2.0))
        padded_chi = np.zeros((self.nblocks * self.padded_dims))
        padded_chi[:self.chi_arr.shape[0]] = self.chi_arr
        blocks = []
        for i in range(self.nblocks):
            ident = np.eye(self.padded_dims)
            lo, hi = i * self.padded_dims, (i + 1) * self.padded_dims
            for r in range(3):
                ident *= self.radem_diag[r:r + 1, 0, lo:hi] * nc
                self.ops.cpuFastHadamardTransform2D(ident)
            ident *= padded_chi[lo:hi]
            blocks.append(ident.T[:, :self.xdim[-1]])
        w = np.vstack(blocks)[:self.num_freqs, :]
        if not self.double_precision:
            w = w.astype(np.float32)
        self.precomputed_weights = np.ascontiguousarray(w)

    def gradient_x(self, input_x, sequence_length=None):
        if self.precomputed_weights is None:
            self.precompute_weights()
        xin = np.ascontiguousarray(input_x.astype(np.float64 if self.double_precision else np.float32))
        nl = int(self.ard_position_key.max()) + 1
        out = np.zeros((xin.shape[0], self.num_rffs), np.float64)
        grad = np.zeros((xin.shape[0], self.num_rffs, nl), np.float64)
        self.ops.cpuMiniARDGrad(xin, out, self.precomputed_weights, self.ard_position_key,
                                self.full_ard_weights, grad, self.fit_intercept)
        if self.fit_intercept:
            out[:, 0] = 1.
            grad[:, 0, :] = 0.
        return out, grad


class OracleDataset:
    """In-memory chunk generator: data_handling/online_data_handling.py:54-94
    (y is standardised per chunk with the stored mean/std, :66-68)."""

    def __init__(self, x, y, seqlen=None, chunk_size=2000, normalize_y=True):
        self.x, self.y, self.seqlen = x, y, seqlen
        self.chunk_size = chunk_size
        if normalize_y:
            self.y_mean, self.y_std = float(y.mean()), float(y.std())
        else:
            self.y_mean, self.y_std = 0.0, 1.0

    def get_ndatapoints(self):
        return self.x.shape[0]

    def get_chunked_data(self):
        for i in range(0, self.x.shape[0], self.chunk_size):
            j = min(i + self.chunk_size, self.x.shape[0])
            yc = self.y[i:j].astype(np.float64)
            yc -= self.y_mean
            yc /= self.y_std
            yield self.x[i:j, ...], yc, None if self.seqlen is None else self.seqlen[i:j]

    def get_chunked_x_data(self):
        for i in range(0, self.x.shape[0], self.chunk_size):
            j = min(i + self.chunk_size, self.x.shape[0])
            yield self.x[i:j, ...], None if self.seqlen is None else self.seqlen[i:j]


def calc_zty(dataset, kernel):
    """scoring_toolkit/exact_nmll_calcs.py:13-39."""
    zty = np.zeros((kernel.get_num_rffs()))
    yty = 0.0
    for xin, yin, ldata in dataset.get_chunked_data():
        z = kernel.transform_x(xin, ldata)
        zty += z.T @ yin
        yty += float((yin ** 2).sum())
    return zty, yty


def matvec(dataset, kernel, vec, out):
    """fitting_toolkit/cg_tools.py:173-200 (regression branch)."""
    out[:] = 0
    for x, lengths in dataset.get_chunked_x_data():
        z = kernel.transform_x(x, lengths)
        out += z.T @ (z @ vec)
    out += kernel.get_lambda() ** 2 * vec


def cg_fit(dataset, kernel, preconditioner, resid, maxiter=200, tol=1e-4, trace=None, nmll_settings=False):
    """fitting_toolkit/cg_tools.py:203-302 (CPU_ConjugateGrad.fit, nmll_settings=False).
    Mirrors the lagging ``err`` (computed from the *current* column after the
    next one has been written, :265).  ``trace`` (a dict) receives per-iteration
    x_k / alpha / beta copies for fixtures."""
    converged = False
    target = resid[:, 0, :].copy()
    init_norms = np.linalg.norm(target, axis=0)
    m, k = resid.shape[0], resid.shape[2]
    z_k = np.zeros((m, 2, k))
    p_k = np.zeros((m, 2, k))
    alpha, beta = np.zeros(k), np.zeros(k)
    losses, alphas, betas = [], [], []
    x_k = np.zeros((m, k))
    w = x_k.copy()
    if preconditioner is None:
        z_k[:, 0, :] = resid[:, 0, :]
    else:
        z_k[:, 0, :] = preconditioner.batch_matvec(resid[:, 0, :])
    p_k[:, 0, :] = z_k[:, 0, :]
    nxt, cur = 1, 0
    niter = 0
    for niter in range(maxiter):
        matvec(dataset, kernel, p_k[:, cur, :], w)
        alpha[:] = (resid[:, cur, :] * z_k[:, cur, :]).sum(axis=0) / \
            (p_k[:, cur, :] * w).sum(axis=0)
        x_k += alpha[None, :] * p_k[:, cur, :]
        resid[:, nxt, :] = resid[:, cur, :] - alpha[None, :] * w
        err = np.linalg.norm(resid[:, cur, :], axis=0) / init_norms
        if preconditioner is None:
            z_k[:, nxt, :] = resid[:, nxt, :]
        else:
            z_k[:, nxt, :] = preconditioner.batch_matvec(resid[:, nxt, :])
        beta[:] = (resid[:, nxt, :] * z_k[:, nxt, :]).sum(axis=0) / \
            (resid[:, cur, :] * z_k[:, cur, :]).sum(axis=0)
        p_k[:, nxt, :] = z_k[:, nxt, :] + beta[None, :] * p_k[:, cur, :]
        losses.append(float(err[0]))
        alphas.append(alpha.copy())
        betas.append(beta.copy())
        if trace is not None:
            trace.setdefault("x_k", []).append(x_k.copy())
            trace.setdefault("alpha", []).append(alpha.copy())
            trace.setdefault("beta", []).append(beta.copy())
        nxt, cur = abs(nxt - 1), abs(cur - 1)
        if err.max() < tol:
            converged = True
            break
    if nmll_settings:        # cg_tools.py:296-299
        return x_k, np.stack(alphas)[:, 1:], np.stack(betas)[:, 1:]
    if x_k.shape[1] > 1:
        return x_k, converged, niter + 1, losses
    return x_k[:, 0], converged, niter + 1, losses


def cg_fit_lib_internal(kernel, dataset, cg_tol=1e-4, max_iter=500, preconditioner=None,
                        trace=None):
    """fitting_toolkit/cg_fitting_toolkit.py:18-70."""
    resid = np.zeros((kernel.get_num_rffs(), 2, 1))
    if preconditioner is None:
        zty, _ = calc_zty(dataset, kernel)
    else:
        zty = preconditioner.get_zty()
    resid[:, 0, :] = zty[:, None] / dataset.get_ndatapoints()
    weights, converged, n_iter, losses = cg_fit(dataset, kernel, preconditioner, resid,
                                                max_iter, cg_tol, trace)
    weights *= dataset.get_ndatapoints()
    return weights, n_iter, losses, converged


class OracleSRHTCompressor:
    """kernels/srht_compressor.py:37-97 (double precision, as the preconditioner uses it)."""

    def __init__(self, compression_size, input_size, random_seed=123, ops=None):
        if compression_size >= input_size or compression_size <= 1:
            raise RuntimeError("The compression size should be < the number of rffs and > 1.")
        self.ops = ops if ops is not None else Oracle()
        self.compression_size, self.input_size = compression_size, input_size
        self.padded_dims = padded_dims(input_size)
        self.radem, self.col_sampler = draw_srht_params(compression_size, input_size, random_seed)
        self.truncated_sampler = self.col_sampler[:compression_size]

    def transform_x(self, features, no_compression=False):
        if features.shape[1] != self.input_size or features.ndim != 2:
            raise RuntimeError("Input with unexpected size passed to a compressor module.")
        if features.shape[1] < self.padded_dims:
            xf = np.zeros((features.shape[0], self.padded_dims), np.float64)
            xf[:, :features.shape[1]] = features
        else:
            xf = features.astype(np.float64)
        self.ops.cpuSRHT(xf, self.radem)
        if no_compression:
            return xf[:, self.col_sampler]
        return xf[:, self.truncated_sampler]


class OracleRandNysPreconditioner:
    """preconditioners/rand_nys_preconditioners.py:18-72 over
    preconditioners/rand_nys_constructors.py:96-123 (single_pass_srht_zty),
    :18-36 (single_pass_gauss), :221-296 (initialize_srht), :127-218
    (initialize_srht_multipass).  Regression only."""

    def __init__(self, kernel, dataset, max_rank, random_state=123, method="srht"):
        if method not in ["srht_2", "srht_3", "srht"]:
            raise RuntimeError("Unknown method supplied for tuning preconditioner construction.")
        m = kernel.get_num_rffs()
        acc = np.zeros((max_rank, m))
        zty = np.zeros((m))
        yty = 0.0
        comp = OracleSRHTCompressor(max_rank, m, random_seed=random_state, ops=kernel.ops)
        for xin, yin, ldata in dataset.get_chunked_data():
            z = kernel.transform_x(xin, ldata)
            zty += z.T @ yin
            yty += yin.T @ yin
            acc += comp.transform_x(z).T @ z
        if method == "srht":
            c_mat = comp.transform_x(acc)
            _, s1, v1 = np.linalg.svd(c_mat, full_matrices=False)
            mask = s1 < 1e-14
            s1 = 1 / np.sqrt(s1.clip(min=1e-14))
            s1[mask] = 0
            acc = acc.T @ v1.T @ (s1[:, None] * v1)
            u_mat, s_mat, _ = np.linalg.svd(acc, full_matrices=False)
            s_mat = s_mat ** 2
        else:
            import scipy.linalg
            n_passes = int(method.split("_")[1])
            acc = acc.T
            for _ in range(n_passes - 1):
                q_mat, _r = np.linalg.qr(acc)
                acc[:] = 0.0
                for xd, ld in dataset.get_chunked_x_data():
                    z = kernel.transform_x(xd, ld)
                    acc += z.T @ (z @ q_mat)
            norm = float(np.sqrt((acc ** 2).sum()))
            shift = np.spacing(norm)
            acc += shift * q_mat
            q_mat = q_mat.T @ acc
            q_mat = np.linalg.cholesky(q_mat)
            acc = scipy.linalg.solve_triangular(q_mat, acc.T, overwrite_b=True, lower=True).T
            u_mat, s_mat, _ = np.linalg.svd(acc, full_matrices=False)
            s_mat = (s_mat ** 2 - shift).clip(min=0)
        self.u_mat, self.eig, self.z_trans_y, self.y_trans_y = u_mat, s_mat, zty, yty
        lambda_ = kernel.get_lambda()
        min_eig = self.eig.min()
        self.eig = self.eig + lambda_ ** 2
        self.inv_eig = self.eig.copy()
        mask = self.inv_eig > 1e-14
        self.inv_eig[mask] = 1 / self.inv_eig[mask]
        self.inv_eig[mask == False] = 0.0  # noqa: E712
        self.achieved_ratio = min_eig / lambda_ ** 2
        self.prefactor = float(min_eig + lambda_ ** 2)

    def batch_matvec(self, xvec):
        xprod = self.u_mat.T @ xvec
        xprod1 = self.u_mat @ (self.inv_eig[:, None] * self.prefactor * xprod)
        xprod2 = xvec - (self.u_mat @ xprod)
        return xprod2 + xprod1

    def get_zty(self):
        return self.z_trans_y

    def get_yty(self):
        return float(self.y_trans_y)

    def get_logdet(self):
        """preconditioners/rand_nys_preconditioners.py:96-102."""
        logdet = 1 + (self.eig - self.prefactor) / self.prefactor
        return float(np.log(logdet.clip(min=1e-12)).sum())

    def matvec_for_sampling(self, xvec):
        """preconditioners/rand_nys_preconditioners.py:105-119."""
        eigvals = np.sqrt(self.eig.clip(min=0))
        prefactor = np.sqrt(1 / self.prefactor)
        xprod = self.u_mat.T @ xvec
        xprod1 = self.u_mat @ (eigvals[:, None] * prefactor * xprod)
        xprod2 = xvec - (self.u_mat @ xprod)
        return xprod1 + xprod2


# ---------------------------------------------------------------------------------------
# NMLL: exact, gradient, approximate (stochastic Lanczos quadrature)
# ---------------------------------------------------------------------------------------
def optimize_alpha_beta(lambda_, nll_terms, ndatapoints, nrffs, beta_max=10., beta_min=0.1):
    """scoring_toolkit/alpha_beta_optimizer.py:13-39."""
    beta = np.sqrt(2 * nll_terms[0] / (ndatapoints * lambda_ ** 2))
    beta = max(min(beta, beta_max), beta_min)
    score = nll_terms[0] / (beta * lambda_) ** 2 + (ndatapoints - nrffs) * np.log(lambda_)
    score += nll_terms[1] + ndatapoints * np.log(beta)
    return score + 0.5 * ndatapoints * np.log(2 * np.pi), beta


def generate_normal_probes(nsamples, num_rffs, random_seed=123, preconditioner=None):
    """scoring_toolkit/probe_generators.py:54-75."""
    rng = np.random.default_rng(random_seed)
    probes = rng.standard_normal(size=(num_rffs, nsamples))
    if preconditioner is not None:
        probes = preconditioner.matvec_for_sampling(probes)
    return probes


def estimate_logdet(alphas, betas, num_rffs, preconditioner=None):
    """scoring_toolkit/approximate_nmll_calcs.py:12-50."""
    from scipy.linalg import eigh_tridiagonal
    mat_diag = 1 / alphas
    mat_diag[1:, :] += betas[:-1, :] / alphas[:-1, :]
    upper_diag = np.sqrt(betas) / alphas
    logdets = np.zeros((mat_diag.shape[1]))
    for i in range(mat_diag.shape[1]):
        eigvals, eigvecs = eigh_tridiagonal(mat_diag[:, i], upper_diag[:-1, i], lapack_driver="stev")
        weights = eigvecs[0, :] ** 2
        logdets[i] += (weights * np.log(eigvals)).sum()
    logdets = num_rffs * logdets.sum() / alphas.shape[1]
    if preconditioner is not None:
        logdets += preconditioner.get_logdet()
    return float(logdets)


def approximate_nmll(kernel, dataset, preconditioner, nsamples=25, nmll_iter=500, nmll_tol=1e-6,
                     random_seed=123, details=None):
    """xgp_regression.py:338-367 (the preconditioner is the caller's)."""
    m = kernel.get_num_rffs()
    n = dataset.get_ndatapoints()
    resid = np.zeros((m, 2, nsamples + 1))
    probes = generate_normal_probes(nsamples, m, random_seed, preconditioner)
    zty, yty = preconditioner.get_zty(), preconditioner.get_yty()
    resid[:, 0, 0] = zty / n
    resid[:, 0, 1:] = probes
    x_k, alphas, betas = cg_fit(dataset, kernel, preconditioner, resid, nmll_iter, nmll_tol, nmll_settings=True)
    x_k[:, 0] *= n
    logdet = estimate_logdet(alphas, betas, m, preconditioner)
    nll1 = float(0.5 * (yty - zty.T @ x_k[:, 0]))
    negloglik, _ = optimize_alpha_beta(kernel.get_lambda(), np.array([nll1, 0.5 * logdet]), n, m)
    if details is not None:
        details.update(alphas=alphas, betas=betas, logdet=logdet, weights=x_k[:, 0], probes=probes)
    return float(negloglik)


def exact_nmll(kernel, dataset):
    """xgp_regression.py:152-205 with scoring_toolkit/exact_nmll_calcs.py:42-110."""
    from scipy.linalg import cho_solve
    m = kernel.get_num_rffs()
    ztz, zty, yty = np.zeros((m, m)), np.zeros(m), 0.0
    for xin, yin, ldata in dataset.get_chunked_data():
        z = kernel.transform_x(xin, ldata)
        zty += z.T @ yin
        ztz += z.T @ z
        yty += float(yin.T @ yin)
    ztz.flat[::m + 1] += kernel.get_lambda() ** 2
    chol = np.linalg.cholesky(ztz)
    weights = cho_solve((chol, True), zty)
    nll1 = float(0.5 * (yty - zty.T @ weights))
    nll2 = float(np.log(np.diag(chol)).sum())
    negloglik, _ = optimize_alpha_beta(kernel.get_lambda(), np.array([nll1, nll2]), dataset.get_ndatapoints(), m)
    return float(negloglik)


def exact_nmll_gradient(kernel, dataset):
    """xgp_regression.py:209-260 with scoring_toolkit/nmll_gradient_tools.py:12-162 (subsample = 1).
    ``kernel.gradient_x`` supplies (Z, dZ/dsigma) (kernels/kernel_baseclass.py:328-361)."""
    from scipy.linalg import cho_solve, solve_triangular
    m = kernel.get_num_rffs()
    hparams = np.asarray(kernel.hyperparams, dtype=np.float64)
    nk = hparams.shape[0] - 1
    ztz, zty, yty = np.zeros((m, m)), np.zeros(m), 0.0
    dzty, inner = np.zeros((m, nk)), np.zeros((m, m, nk))
    n = 0
    for xin, yin, ldata in dataset.get_chunked_data():
        z, dz = kernel.gradient_x(xin, ldata)
        zty += z.T @ yin
        ztz += z.T @ z
        yty += float(yin.T @ yin)
        n += z.shape[0]
        for i in range(nk):
            dzty[:, i] += dz[:, :, i].T @ yin
            inner[:, :, i] += dz[:, :, i].T @ z
    inner += np.transpose(inner, (1, 0, 2))
    lam = hparams[0]
    ztz.flat[::m + 1] += lam ** 2
    chol = np.linalg.cholesky(ztz)
    weights = cho_solve((chol, True), zty)
    chol_inv = solve_triangular(chol, np.eye(m), lower=True)
    nll1 = float(0.5 * (yty - zty.T @ weights))
    nll2 = float(np.log(np.diag(chol)).sum())
    negloglik, beta = optimize_alpha_beta(lam, np.array([nll1, nll2]), float(n), float(m))
    grad = np.zeros(hparams.shape[0])
    alpha = lam * beta
    g0 = (1 / (beta ** 2 * lam ** 3)) * ((zty.T @ weights) - yty)
    g0 += (1 / (beta ** 2 * lam)) * (weights.T @ weights)
    g0 += (n - m) / lam
    g0 += lam * (chol_inv ** 2).sum()
    grad[0] = float(g0)
    for i in range(nk):
        trace_term = cho_solve((chol, True), inner[:, :, i])
        g = -2 * (weights.T @ dzty[:, i])
        g += weights.T @ (inner[:, :, i] @ weights)
        g *= 0.5 / alpha ** 2
        g += 0.5 * trace_term.trace()
        grad[i + 1] = float(g)
    grad *= hparams
    return float(negloglik), grad


# ---------------------------------------------------------------------------------------
# Classification (fitting_toolkit/nonlinear_cg_toolkit.py, xgp_classification.py)
# ---------------------------------------------------------------------------------------
class OracleClassificationDataset(OracleDataset):
    """data_handling/dataset_builder.py:68-117, :182-188: integer labels, no y normalisation."""

    def __init__(self, x, y, seqlen=None, chunk_size=2000):
        if not np.issubdtype(y.dtype, np.integer):
            raise RuntimeError("For classification, ydata must be an array of integers.")
        if y.min() != 0:
            raise RuntimeError("For classification, there must be a zero category.")
        super().__init__(x, y, seqlen, chunk_size, normalize_y=False)
        self.y = y
        self.n_classes = int(y.max()) + 1

    def get_n_classes(self):
        return self.n_classes

    def get_chunked_data(self):
        n = self.x.shape[0]
        for i in range(0, n, self.chunk_size):
            j = min(i + self.chunk_size, n)
            yield self.x[i:j], self.y[i:j], None if self.seqlen is None else self.seqlen[i:j]


def _softmax_ref(pred):
    """the reference's softmax: base 2.71828, not e (nonlinear_cg_toolkit.py:252-254)."""
    pred = pred - pred.max(axis=1)[:, None]
    pred = 2.71828 ** pred
    return pred / pred.sum(axis=1)[:, None]


def classification_cost(dataset, kernel, wvec):
    """nonlinear_cg_toolkit.py:231-275 -> (grad, loss)."""
    lam = kernel.get_lambda()
    grad = np.zeros(wvec.shape)
    grad[1:, :] += lam ** 2 * wvec[1:, :]
    loss = 0.5 * lam ** 2 * (wvec ** 2)[1:, :].sum()
    for xd, yd, ld in dataset.get_chunked_data():
        z = kernel.transform_x(xd, ld)
        yd = yd.astype(np.int32)
        pred = _softmax_ref(z @ wvec)
        loss -= float(np.log(pred.clip(min=1e-16))[np.arange(pred.shape[0]), yd].sum())
        for k in range(wvec.shape[1]):
            grad[:, k] += ((pred[:, k] - (yd == k).astype(np.float64))[:, None] * z).sum(axis=0)
    return grad, float(loss)


def fit_classifier(dataset, kernel, preconditioner=None, max_iter=500, tol=1e-4):
    """nonlinear_cg_toolkit.py:72-226 -> (weights, n_iter, losses)."""
    state = {"last_grad": None, "last_sd": None, "n_iter": 0}

    def cost(w):
        return classification_cost(dataset, kernel, w)

    def update(grad, wvec, loss, previous_loss):
        sd = preconditioner.batch_matvec(grad) if preconditioner is not None else grad
        if state["last_grad"] is not None:
            pr = (sd * (grad - state["last_grad"])).sum() / (state["last_grad"] * state["last_sd"]).sum()
            pr = max(0., float(pr))
            correction = pr * state["last_sd"]
            state["last_grad"], state["last_sd"] = grad.copy(), sd.copy()
            sd += correction           # in place: aliases grad when there is no preconditioner (:136, :148)
        else:
            state["last_grad"], state["last_sd"] = grad.copy(), sd.copy()
        sd = -sd
        a0p = (grad * sd).sum()
        a_init = 1 if previous_loss is None else 2 * (loss - previous_loss) / a0p
        w_full = wvec + a_init * sd
        g_full, l_full = cost(w_full)
        if state["n_iter"] >= 10 and np.abs(np.abs(l_full - loss) / loss) > tol \
                and l_full < (loss + a_init * 1e-4 * a0p):
            return g_full, l_full, w_full
        a_quad = -(a0p * a_init ** 2) / (2 * (l_full - loss - a0p * a_init))
        w_quad = wvec + a_quad * sd
        g_quad, l_quad = cost(w_quad)
        if l_quad < l_full:
            if l_quad < (loss + a_quad * 1e-4 * a0p):
                return g_quad, l_quad, w_quad
        elif l_full < (loss + a_init * 1e-4 * a0p):
            return g_full, l_full, w_full
        cand = [(loss, grad, wvec), (l_full, g_full, w_full), (l_quad, g_quad, w_quad)]
        a_max = a_quad if l_quad < l_full else a_init
        rfactor = 0.5
        for _ in range(10):
            a = rfactor * a_max
            w_c = wvec + a * sd
            g_c, l_c = cost(w_c)
            if l_c < (loss + a * 1e-4 * a0p):
                return g_c, l_c, w_c
            cand.append((l_c, g_c, w_c))
            rfactor *= 0.5
        best = int(np.argmin([c[0] for c in cand]))
        return cand[best][1], cand[best][0], cand[best][2]

    wvec = np.zeros((kernel.get_num_rffs(), dataset.get_n_classes()))
    grad, loss = cost(wvec)
    losses = [loss]
    last_alpha = None
    while state["n_iter"] < max_iter:
        grad, loss, wvec = update(grad, wvec, loss, last_alpha)
        losses.append(loss)
        if np.abs(np.abs(losses[-1] - losses[-2]) / losses[-2]) < tol:
            break
        state["n_iter"] += 1
        last_alpha = losses[state["n_iter"] - 1]
    return wvec, state["n_iter"], losses


def predict_proba(kernel, weights, input_x, sequence_lengths=None, gamma=None):
    """xgp_classification.py:59-109."""
    z = kernel.transform_x(input_x, sequence_lengths)
    pred = z @ weights
    if gamma is not None:
        pred = pred + gamma[None, :]
    return _softmax_ref(pred)


# ---------------------------------------------------------------------------------------
# Preconditioner rank selection (model_baseclass.py:376-480, rand_nys_constructors.py:60-93, :301-357)
# ---------------------------------------------------------------------------------------
def srht_ratio_check(dataset, rank, kernel, random_state=123, sample_frac=0.1):
    """rand_nys_constructors.py:301-357 with the sampled pass of :60-93."""
    m = kernel.get_num_rffs()
    acc = np.zeros((rank, m))
    comp = OracleSRHTCompressor(rank, m, random_seed=random_state, ops=kernel.ops)
    rng = np.random.default_rng(random_state)
    for xd, ld in dataset.get_chunked_x_data():
        cutoff = max(int(sample_frac * float(xd.shape[0])), 1)
        idx = rng.permutation(xd.shape[0])[:cutoff]
        z = kernel.transform_x(xd[idx, ...], None if ld is None else ld[idx])
        acc += comp.transform_x(z).T @ z
    c_mat = comp.transform_x(acc)
    _, s1, v1 = np.linalg.svd(c_mat, full_matrices=False)
    mask = s1 < 1e-14
    s1 = 1 / np.sqrt(s1.clip(min=1e-14))
    s1[mask] = 0
    acc = acc.T @ v1.T @ (s1[:, None] * v1)
    _, s_mat, _ = np.linalg.svd(acc, full_matrices=False)
    return s_mat ** 2


def check_rank_ratio(kernel, dataset, sample_frac=0.1, max_rank=512, random_seed=123):
    """model_baseclass.py:438-480 (kernels of at most 8192 random features)."""
    s_mat = srht_ratio_check(dataset, max_rank, kernel, random_seed, sample_frac)
    return float(s_mat.min() / kernel.get_lambda() ** 2) / sample_frac


def autoselect_rank(kernel, dataset, min_rank=512, max_rank=3000, increment_size=512, always_use_srht2=False,
                    ratio_target=30., random_seed=123):
    """model_baseclass.py:376-436 -> (rank, method) the preconditioner is then built with."""
    sample_frac, method, ratio, rank = 0.2, "srht", np.inf, min_rank
    m = kernel.get_num_rffs()
    if rank >= m:
        rank = m - 1
        ratio = 0.5 * ratio_target
    if dataset.get_ndatapoints() < 5000:
        sample_frac = 1
    while ratio > ratio_target and rank < max_rank:
        ratio = check_rank_ratio(kernel, dataset, sample_frac, rank, random_seed)
        if ratio > ratio_target:
            if (rank + increment_size) < max_rank and (rank + increment_size) < m:
                rank += increment_size
            else:
                rank = max_rank
                if rank > m:
                    rank = m - 1
                method = "srht_2"
                break
    if always_use_srht2:
        method = "srht_2"
    return rank, method
