"""Host-side counterparts of the reference's kernel objects for the hot path.

Mirrors (paths relative to /root/reference/src/xGPR/kernels/):
  * ``KernelBaseclass`` transform_x / transform_x_y          kernel_baseclass.py:269-324
  * ``SORFKernelBaseclass`` (RBF, Matern, Cauchy)             basic_kernels/sorf_kernel_baseclass.py:36-126,
                                                              matern.py:26-58, cauchy.py:21-45
  * ``ConvKernelBaseclass`` (Conv1d*/Graph* RBF/Matern/Cauchy) convolution_kernels/conv_kernel_baseclass.py:41-147
  * ``SRHTCompressor``                                        srht_compressor.py:37-97

The random draws (Rademacher diagonals, chi / chi-square / exponential samples, the SRHT
column permutation) are made on the host with exactly the numpy / scipy calls the reference
makes -- they are *inputs* of the GPU path and must be bit-identical (tests/golden/g6_draws.npz).
Everything that touches datapoints runs in libxgpr_hip.so on the device.
"""
from math import ceil

import numpy as np
import torch
from scipy.stats import chi as _chi

from . import xgpr_hip_rfgen_ext as ext


def padded_dims(width):
    return 2 ** ceil(np.log2(max(width, 2)))


def scale_input(x, sigma, out_dtype=torch.float32):
    """``input_x *= self.hyperparams[1]`` (sorf_kernel_baseclass.py:117) on a private copy.
    hyperparams[1] is an np.float64 *scalar*: under numpy >= 2 (NEP 50, the version the golden
    vectors were produced with) the product is formed in float64 and rounded back to the
    array's dtype."""
    return (x.to(torch.float64) * float(sigma)).to(out_dtype).contiguous()


class KernelBase:
    """State shared by all hot-path kernels (kernel_baseclass.py:49-99)."""

    def __init__(self, num_rffs, xdim, kernel_spec_parms=None, device="cuda"):
        kernel_spec_parms = kernel_spec_parms or {}
        if num_rffs < 2:
            raise RuntimeError("num_rffs should always be >= 2.")
        if not (num_rffs / 2).is_integer():
            raise RuntimeError("For sine-cosine kernels (e.g. matern, rbf) the number of random "
                               "fourier features must be an integer multiple of two.")
        self.num_freqs = int(num_rffs / 2)
        self.num_rffs = int(num_rffs)
        self.kernel_spec_parms = dict(kernel_spec_parms)
        self.random_seed = 123           # subclasses overwrite it with the seed they draw with
        self.fit_intercept = kernel_spec_parms.get("intercept", True) is not False
        self._xdim = tuple(xdim)
        self.hyperparams = np.ones((2))
        self.device = device
        self.double_precision = False

    # ---- hyperparameters (kernel_baseclass.py:219-266)
    def get_hyperparams(self, logspace=True):
        return np.log(self.hyperparams) if logspace else self.hyperparams

    def set_hyperparams(self, hyperparams, logspace=True):
        self.hyperparams = np.exp(hyperparams) if logspace else np.asarray(hyperparams, dtype=np.float64)

    def get_lambda(self):
        return self.hyperparams[0]

    def get_num_rffs(self):
        return self.num_rffs

    def _to_device(self, radem, chi_arr):
        self.radem_diag = torch.from_numpy(np.ascontiguousarray(radem)).to(self.device)
        self.chi_arr = torch.from_numpy(np.ascontiguousarray(chi_arr)).to(self.device)

    def _as_device(self, input_x):
        if isinstance(input_x, np.ndarray):
            input_x = torch.from_numpy(np.ascontiguousarray(input_x))
        return input_x.to(self.device)

    def _as_device_f32(self, input_x):
        """the private float32 copy every transform starts from (kernel_baseclass.py:274-288): float64 inputs
        are rounded to float32 BEFORE sigma is applied, as the reference does"""
        return self._as_device(input_x).to(torch.float32)

    def transform_x(self, input_x, sequence_length=None):
        """kernel_baseclass.py:269-299: private float32 copy -> kernel_specific_transform ->
        ``xtrans[:, 0] = 1`` when fitting an intercept.  Returns a float64 device tensor."""
        xin = scale_input(self._as_device_f32(input_x), self.hyperparams[1])
        xtrans = self.kernel_specific_transform(xin, sequence_length)
        if self.fit_intercept:
            xtrans[:, 0] = 1.
        return xtrans

    def gradient_x(self, input_x, sequence_length=None):
        """kernel_baseclass.py:328-361: features and d(features)/d(sigma) from the *unscaled* input
        (the gradient operators apply sigma themselves).  Returns float64 [n, M], [n, M, 1]."""
        xin = self._as_device_f32(input_x).to(torch.float32, copy=True).contiguous()
        xtrans, xgrad = self.kernel_specific_gradient(xin, sequence_length)
        if self.fit_intercept:
            xtrans[:, 0] = 1.
            xgrad[:, 0, :] = 0.
        return xtrans, xgrad

    def gradient_x_y(self, input_x, input_y, sequence_length=None):
        """kernel_baseclass.py:364-377."""
        xtrans, dz_dsigma = self.gradient_x(input_x, sequence_length)
        if isinstance(input_y, np.ndarray):
            input_y = torch.from_numpy(input_y)
        return xtrans, dz_dsigma, input_y.to(self.device, torch.float64)

    def transform_x_y(self, input_x, input_y, sequence_length=None):
        """kernel_baseclass.py:303-324 (regression branch)."""
        xtrans = self.transform_x(input_x, sequence_length)
        if isinstance(input_y, np.ndarray):
            input_y = torch.from_numpy(input_y)
        return xtrans, input_y.to(self.device, torch.float64)


class SORFKernel(KernelBase):
    """RBF / Matern / Cauchy on fixed-length vectors."""

    def __init__(self, kernel_choice, xdim, num_rffs, random_seed=123, device="cuda",
                 kernel_spec_parms=None):
        kernel_spec_parms = kernel_spec_parms or {}
        super().__init__(num_rffs, xdim, kernel_spec_parms, device)
        self.random_seed = random_seed
        if len(xdim) != 2:
            raise ValueError("The dimensionality of the input is inappropriate for "
                             "the kernel you have selected.")
        self.kernel_choice = kernel_choice
        pdims = padded_dims(xdim[-1])
        radem_array = np.asarray([-1, 1], dtype=np.int8)
        rng = np.random.default_rng(random_seed)
        nblocks = ceil(self.num_freqs / pdims) if pdims < self.num_freqs else 1
        radem = rng.choice(radem_array, size=(3, 1, nblocks * pdims), replace=True)
        chi_arr = _chi.rvs(df=pdims, size=self.num_freqs, random_state=random_seed).astype(np.float32)
        chi_arr = _rescale_chi(kernel_choice, chi_arr, random_seed, kernel_spec_parms, self)
        self._to_device(radem, chi_arr)

    def kernel_specific_transform(self, input_x, sequence_length=None):
        """sorf_kernel_baseclass.py:104-126; ``input_x`` is already sigma-scaled here."""
        output_x = torch.empty((input_x.shape[0], self.num_rffs), dtype=torch.float64, device=self.device)
        ext.hipRBFFeatureGen(input_x, output_x, self.radem_diag, self.chi_arr, self.fit_intercept)
        return output_x

    def kernel_specific_gradient(self, input_x, sequence_length=None):
        """sorf_kernel_baseclass.py:136-162."""
        n = input_x.shape[0]
        output_x = torch.zeros((n, self.num_rffs), dtype=torch.float64, device=self.device)
        dz_dsigma = torch.zeros((n, self.num_rffs, 1), dtype=torch.float64, device=self.device)
        ext.hipRBFGrad(input_x, output_x, dz_dsigma, self.radem_diag, self.chi_arr,
                       float(self.hyperparams[1]), self.fit_intercept)
        return output_x, dz_dsigma

    # ---- fused per-shard reductions (what the reference does chunk by chunk with a
    # materialised Z: fitting_toolkit/cg_tools.py:189-191, scoring_toolkit/exact_nmll_calcs.py:35-37)
    supports_fused = True

    def ztz_matvec(self, x_scaled, vec, out, workspace=None, masks_packed=False):
        ext.hipZtZMatvec(x_scaled, self.radem_diag, self.chi_arr, vec, out, self.fit_intercept, workspace, masks_packed)

    def zty(self, x_scaled, y, out, workspace=None):
        ext.hipZtY(x_scaled, self.radem_diag, self.chi_arr, y, out, self.fit_intercept, workspace)

    # ---- resident feature cache: keep the shard's Z in HBM as float32 (cos, sin) pairs and stream
    # it on every CG iteration instead of regenerating it (an option the 288 GB of HBM3E allow;
    # the reference cannot hold Z and regenerates it, cg_tools.py:189-191)
    def cache_ok(self):
        """k = 1 streaming kernel up to num_freqs = 16384 (a workgroup holds all tiles of a datapoint: one tile
        per wave up to 8192, two beyond); past that the resident cache is applied through the two block
        contractions with one column."""
        return self.fused_ok() and (self.num_freqs <= 16384 or self.block_ok())

    def cache_pays(self):
        """Whether streaming the resident cache beats regenerating the features in a k = 1 solve: the launcher's own
        predicate (xgpr_ztz_matvec_plan).  On the three-wave single-pass kernel regenerating a 1024-frequency tile takes
        ~1.30 ns (cfg3: 5.19 ms per 1e6 rows) while the cache streams it in ~1.33 ns (32.8 GB at 6.2 TB/s: 5.32 ms) --
        regenerating wins or ties (cfg2: 0.32 / 0.33 ms) and leaves the HBM free.  Every other plan loses to the stream:
        the two-wave kernel (one tile per datapoint at padded width >= 128, seven tiles), the two feature passes (eight
        tiles per datapoint, or more than 8192 frequencies; cfg5's share: 9.1 / 6.1 ms) and the wide transforms of padded
        width 2048 / 4096 (cross-wave stages: slower per tile than the stream).  bench.py reports both modes
        (`cached_z_mode`)."""
        return ext.ztz_matvec_plan(self._xdim[-1], self.num_freqs) != 1 or padded_dims(self._xdim[-1]) > 1024

    def build_feature_cache(self, dataset):
        x_scaled = dataset.scaled_x(self.hyperparams[1])
        zc = torch.empty((x_scaled.shape[0], self.num_rffs), dtype=torch.float32, device=self.device)
        ext.hipRBFFeatureCache(x_scaled, zc, self.radem_diag, self.chi_arr)
        return zc

    def ztz_matvec_cached(self, zcache, vec, out, workspace):
        if self.num_freqs <= 16384:
            ext.hipZCacheMatvec(zcache, vec, out, self.fit_intercept, workspace)
            return
        need = block_workspace_bytes(zcache.shape[0], self.num_rffs, 1)
        if getattr(self, "_cache_bws", None) is None or self._cache_bws.numel() < need:
            self._cache_bws = torch.empty(need, dtype=torch.uint8, device=zcache.device)
        ext.hipZCacheBlockMatvec(zcache, vec[:, None], out[:, None], self.fit_intercept, self._cache_bws)

    # ---- block of right-hand sides (approximate-NMLL probes, k = 26): float64 matrix cores over
    # the float32 cache, either the resident one or a window of rows regenerated into scratch
    def block_ok(self):
        return padded_dims(self._xdim[-1]) <= FUSED_MAX_WIDTH and self.num_rffs % 4 == 0

    def fill_feature_cache(self, x_scaled, zcache):
        ext.hipRBFFeatureCache(x_scaled, zcache, self.radem_diag, self.chi_arr)

    def ztz_block_cached(self, zcache, vecs, out, workspace, accumulate=False):
        _block_matvec(zcache, vecs, out, workspace, self.fit_intercept, 0.0, accumulate)

    def row_cache_params(self):
        """(fit_intercept, scale): Z = scale * cache rows with Z[:, 0] = 1 when fit_intercept -- what the operators
        that consume float32 rows directly (hipSRHTSampleRows, hipSketchGemm) are told."""
        return self.fit_intercept, float(np.float32(np.sqrt(1.0 / (self.num_freqs - 0.5 if self.fit_intercept
                                                                   else self.num_freqs))))

    def cache_rows_to_features(self, zrows):
        """float64 feature rows (what transform_x returns) from rows of the float32 cache: the cache holds
        the (cos, sin) values before scaling, the operator widens them and multiplies by its float-typed
        constant (rbf_ops.cpp:68-72)."""
        scale = float(np.float32(np.sqrt(1.0 / (self.num_freqs - 0.5 if self.fit_intercept else self.num_freqs))))
        z = zrows.to(torch.float64) * scale
        if self.fit_intercept:
            z[:, 0] = 1.
        return z

    def fused_ok(self):
        """The fused kernels cover padded width <= 4096 (single pass up to num_freqs = 7168 -- 4096 at padded widths
        2048 / 4096, whose transforms span two / four wave tiles --, the two-pass form beyond, up to 65536)."""
        return padded_dims(self._xdim[-1]) <= FUSED_MAX_WIDTH and self.num_freqs <= 65536

    def workspace_bytes(self):
        return ext.ztz_workspace_bytes(self.num_rffs, self.radem_diag.shape[2])


FUSED_MAX_WIDTH = 4096   # padded input width the wave-tile kernels serve (include/xgpr_hip.h); beyond it: the any-width LDS path

BLOCK_COLS = 32          # right-hand sides per call of the block matvec (include/xgpr_hip.h)


def _block_matvec(zcache, vecs, out, workspace, fit_intercept, scale, accumulate):
    """out[M, k] (+)= Z^T (Z vecs) in column groups of BLOCK_COLS; vecs / out C-contiguous float64."""
    k = vecs.shape[1]
    if k <= BLOCK_COLS:
        ext.hipZCacheBlockMatvec(zcache, vecs, out, fit_intercept, workspace, scale, accumulate)
        return
    for j0 in range(0, k, BLOCK_COLS):
        j1 = min(k, j0 + BLOCK_COLS)
        v = vecs[:, j0:j1].contiguous()
        o = out[:, j0:j1].contiguous()
        ext.hipZCacheBlockMatvec(zcache, v, o, fit_intercept, workspace, scale, accumulate)
        out[:, j0:j1] = o


def block_workspace_bytes(nrows, num_rffs, k):
    return ext.zcache_block_workspace_bytes(nrows, num_rffs, min(k, BLOCK_COLS))


def _rescale_chi(kernel_choice, chi_arr, random_seed, parms, obj):
    """Matern: matern.py:45-54 (conv twins conv1d_matern.py:58-62, graph_matern.py:51-55);
    Cauchy: cauchy.py:39-41 (conv1d_cauchy.py:52-54, graph_cauchy.py:45-47).  The rescale is
    applied in place to the float32 array, as in the reference."""
    if kernel_choice.endswith("Matern"):
        if "matern_nu" not in parms:
            raise ValueError("Tried to initialize a Matern kernel without supplying nu.")
        obj.matern_nu = parms["matern_nu"]
        if obj.matern_nu < 1 / 2 or obj.matern_nu > 5 / 2:
            raise ValueError("nu must be >= 1/2 and <= 5/2.")
        rng = np.random.default_rng(random_seed)
        chisamples = np.sqrt(rng.chisquare(2 * obj.matern_nu, size=chi_arr.shape[0]) / (obj.matern_nu * 2))
        chi_arr /= chisamples
    elif kernel_choice.endswith("Cauchy"):
        rng = np.random.default_rng(random_seed)
        dstsamples = np.sqrt(rng.exponential(size=chi_arr.shape[0]))
        chi_arr *= dstsamples
    return chi_arr


class ConvSORFKernel(KernelBase):
    """Conv1dRBF / Conv1dMatern / Conv1dCauchy and the Graph* kernels (conv_width = 1,
    graph_rbf.py:42-43)."""

    def __init__(self, kernel_choice, xdim, num_rffs, random_seed=123, device="cuda",
                 kernel_spec_parms=None):
        kernel_spec_parms = kernel_spec_parms or {}
        super().__init__(num_rffs, xdim, kernel_spec_parms, device)
        self.random_seed = random_seed
        if len(xdim) != 3:
            raise RuntimeError("Tried to initialize a Conv1d kernel with a 2d x-"
                               "array! x should be a 3d array for Conv1d.")
        self.kernel_choice = kernel_choice
        if kernel_choice.startswith("Graph"):
            self.conv_width = 1
        else:
            if "conv_width" not in kernel_spec_parms:
                raise ValueError("conv_width must be included as a kernel-specific "
                                 "parameter if using a sequence kernel.")
            self.conv_width = kernel_spec_parms["conv_width"]
        averaging = kernel_spec_parms.get("averaging", "none")
        if averaging not in ("none", "sqrt", "full"):
            raise RuntimeError("Unrecognized value for 'averaging' supplied, "
                               "should be one of 'none', 'sqrt', 'full'.")
        self.scaling_type = {"none": 0, "sqrt": 1, "full": 2}[averaging]
        rng = np.random.default_rng(random_seed)
        pdims = padded_dims(self.conv_width * xdim[2])
        init_calc_freqsize = ceil(self.num_freqs / pdims) * pdims
        radem_array = np.asarray([-1, 1], dtype=np.int8)
        radem = rng.choice(radem_array, size=(3, 1, init_calc_freqsize), replace=True)
        chi_arr = _chi.rvs(df=pdims, size=self.num_freqs, random_state=random_seed).astype(np.float32)
        chi_arr = _rescale_chi(kernel_choice, chi_arr, random_seed, kernel_spec_parms, self)
        self._to_device(radem, chi_arr)

    supports_fused = False

    def fused_ok(self):
        return False

    # ---- resident feature cache.  Convolution features cost K k-mers x a full SORF per sequence, so
    # regenerating them on every CG iteration (what the reference does) is by far the most expensive
    # way to apply Z; here the shard's Z (kernel_baseclass.py:269-299 output, intercept column set)
    # is rounded to float32 once -- entries are sums of float32 cos/sin values, so this adds at most
    # 6e-8 relative per entry -- and streamed from HBM afterwards.
    def cache_ok(self):
        return self.num_freqs <= 16384

    def build_feature_cache(self, dataset):
        # in windows of up to CACHE_BUILD_ROWS sequences rather than the dataset's chunks (1024 sequences in
        # BASELINE configs[3]): one wave per (sequence, tile) runs for as long as its sequence has k-mers, and with 3
        # launch-rounds of waves per chunk the long sequences at the end of a launch leave most of the GPU idle
        # (3.5 ms per 1024 sequences = 2.9e5 sequences/s against 3.5e5 on 8192-sequence launches)
        n = dataset.get_local_ndatapoints()
        zc = torch.empty((n, self.num_rffs), dtype=torch.float32, device=self.device)
        if not hasattr(dataset, "get_xdata"):        # any other dataset class: its own chunks
            lo = 0
            for xdata, ldata in dataset.get_chunked_x_data():
                zc[lo:lo + xdata.shape[0]] = self.transform_x(xdata, ldata).to(torch.float32)
                lo += xdata.shape[0]
            return zc
        xall, lall = dataset.get_xdata(), dataset.get_sequence_lengths()
        step = max(1, min(self.CACHE_BUILD_ROWS, (1 << 30) // (8 * self.num_rffs)))      # float64 temporary <= 1 GiB
        for lo in range(0, n, step):
            hi = min(lo + step, n)
            # (no sequence lengths: transform_x raises the reference's "sequence_length is required" error)
            zc[lo:hi] = self.transform_x(xall[lo:hi], None if lall is None else lall[lo:hi]).to(torch.float32)
        return zc

    CACHE_BUILD_ROWS = 8192

    def ztz_matvec_cached(self, zcache, vec, out, workspace):
        ext.hipZCacheMatvecScaled(zcache, vec, out, 1.0, workspace)

    def block_ok(self):
        return self.num_rffs % 4 == 0

    def ztz_block_cached(self, zcache, vecs, out, workspace, accumulate=False):
        _block_matvec(zcache, vecs, out, workspace, False, 1.0, accumulate)

    def row_cache_params(self):
        """the convolution cache holds complete feature rows (intercept column included): Z = 1.0 * rows"""
        return False, 1.0

    def cache_rows_to_features(self, zrows):
        """the convolution cache holds complete feature rows rounded to float32"""
        return zrows.to(torch.float64)

    def workspace_bytes(self):
        return ext.ztz_workspace_bytes(self.num_rffs, self.radem_diag.shape[2])

    def kernel_specific_gradient(self, input_x, sequence_length):
        """conv_kernel_baseclass.py:157-190."""
        if sequence_length is None:
            raise RuntimeError("sequence_length is required for convolution kernels.")
        if input_x.shape[2] != self._xdim[2]:
            raise RuntimeError("Unexpected input shape supplied.")
        if isinstance(sequence_length, torch.Tensor):
            sequence_length = sequence_length.cpu().numpy()
        slen = np.ascontiguousarray(sequence_length.astype(np.int32, copy=False))
        n = input_x.shape[0]
        xtrans = torch.zeros((n, self.num_rffs), dtype=torch.float64, device=self.device)
        dz_dsigma = torch.zeros((n, self.num_rffs, 1), dtype=torch.float64, device=self.device)
        ext.hipConvGrad(input_x, xtrans, self.radem_diag, self.chi_arr, slen, dz_dsigma,
                        float(self.hyperparams[1]), self.conv_width, self.scaling_type)
        return xtrans, dz_dsigma

    def kernel_specific_transform(self, input_x, sequence_length):
        """conv_kernel_baseclass.py:116-147."""
        if sequence_length is None:
            raise RuntimeError("sequence_length is required for convolution kernels.")
        if input_x.shape[2] != self._xdim[2]:
            raise RuntimeError("Unexpected input shape supplied.")
        if isinstance(sequence_length, torch.Tensor):
            sequence_length = sequence_length.cpu().numpy()
        slen = np.ascontiguousarray(sequence_length.astype(np.int32, copy=False))
        xtrans = torch.zeros((input_x.shape[0], self.num_rffs), dtype=torch.float64, device=self.device)
        ext.hipConv1dFGen(input_x, xtrans, self.radem_diag, self.chi_arr, slen,
                          self.conv_width, self.scaling_type)
        return xtrans


class LinearKernel:
    """kernels/basic_kernels/linear.py: Bayesian linear regression -- the "features" are the input itself
    (rounded to float32 by the private copy of kernel_baseclass.py:274-288, widened to float64), with a
    leading column of ones for the intercept.  No random features, no operator of the library; here so that
    every kernel name of the reference's registry (kernels/__init__.py:21-33) resolves."""
    supports_fused = False
    kernel_choice = "Linear"

    def __init__(self, xdim, num_rffs=None, random_seed=123, device="cuda", kernel_spec_parms=None):
        kernel_spec_parms = kernel_spec_parms or {}
        if len(xdim) > 2:
            raise ValueError("The Linear kernel is only applicable for fixed vector input.")
        self.fit_intercept = kernel_spec_parms.get("intercept", True) is not False
        self.num_rffs = xdim[1] + (1 if self.fit_intercept else 0)
        self._xdim, self.device, self.random_seed = tuple(xdim), device, random_seed
        self.kernel_spec_parms = dict(kernel_spec_parms)
        self.hyperparams = np.ones((1))

    get_hyperparams = KernelBase.get_hyperparams
    set_hyperparams = KernelBase.set_hyperparams
    get_lambda = KernelBase.get_lambda
    get_num_rffs = KernelBase.get_num_rffs
    transform_x_y = KernelBase.transform_x_y
    gradient_x_y = KernelBase.gradient_x_y
    _as_device = KernelBase._as_device
    _as_device_f32 = KernelBase._as_device_f32

    def fused_ok(self):
        return False

    def transform_x(self, input_x, sequence_length=None):
        xin = self._as_device_f32(input_x).to(torch.float32)
        if not self.fit_intercept:
            return xin.to(torch.float64)
        xtrans = torch.zeros((xin.shape[0], xin.shape[1] + 1), dtype=torch.float64, device=self.device)
        xtrans[:, 1:] = xin
        xtrans[:, 0] = 1.
        return xtrans

    def gradient_x(self, input_x, sequence_length=None):
        xtrans = self.transform_x(input_x)
        return xtrans, torch.zeros((xtrans.shape[0], 0, 0), dtype=torch.float64, device=self.device)


class Conv1dTwoLayerKernel(KernelBase):
    """kernels/convolution_kernels/l2_conv1d.py:16-222: a convolution layer with global max-pooling
    (hipConv1dMaxpool: ``init_rffs`` ReLU'd random convolution filters per sequence) whose output feeds an RBF
    kernel (hipRBFFeatureGen / hipRBFGrad); sigma scales the pooled features."""

    def __init__(self, xdim, num_rffs, random_seed=123, device="cuda", kernel_spec_parms=None):
        kernel_spec_parms = kernel_spec_parms or {}
        if "conv_width" not in kernel_spec_parms:
            raise ValueError("conv_width must be included as a kernel-specific "
                             "parameter if using a sequence kernel.")
        if "init_rffs" not in kernel_spec_parms:
            raise ValueError("init_rffs must be included as a kernel-specific "
                             "parameter if using the 2 layer conv1d kernel.")
        if len(xdim) != 3:
            raise RuntimeError("Tried to initialize a Conv1d kernel with a 2d x-"
                               "array! x should be a 3d array for Conv1d.")
        self.init_rffs = kernel_spec_parms["init_rffs"]
        if self.init_rffs % 2 != 0:
            raise RuntimeError("Number of init rffs should be an even number.")
        super().__init__(num_rffs, xdim, kernel_spec_parms, device)
        self.random_seed = random_seed
        self.kernel_choice = "Conv1dTwoLayer"
        rng = np.random.default_rng(random_seed)
        self.conv_width = kernel_spec_parms["conv_width"]
        pdims = padded_dims(self.conv_width * xdim[2])
        init_calc_featsize = ceil(self.init_rffs / pdims) * pdims
        radem_array = np.asarray([-1, 1], dtype=np.int8)
        radem1 = rng.choice(radem_array, size=(3, 1, init_calc_featsize), replace=True)
        chi1 = _chi.rvs(df=pdims, size=self.init_rffs, random_state=random_seed).astype(np.float32)
        pdims2 = padded_dims(self.init_rffs)
        nblocks = ceil(self.num_freqs / pdims2) if pdims2 < self.num_freqs else 1
        radem2 = rng.choice(radem_array, size=(3, 1, nblocks * pdims2), replace=True)
        chi2 = _chi.rvs(df=pdims2, size=self.num_freqs, random_state=random_seed).astype(np.float32)
        self.radem_diag1 = torch.from_numpy(np.ascontiguousarray(radem1)).to(device)
        self.chi_arr1 = torch.from_numpy(chi1).to(device)
        self._to_device(radem2, chi2)            # radem_diag / chi_arr = the second (RBF) layer

    supports_fused = False

    def fused_ok(self):
        return False

    def _first_layer(self, input_x, sequence_length):
        if sequence_length is None:
            raise ValueError("sequence_length is required for convolution kernels.")
        if input_x.shape[2] != self._xdim[2]:
            raise RuntimeError("Unexpected input shape supplied.")
        if isinstance(sequence_length, torch.Tensor):
            sequence_length = sequence_length.cpu().numpy()
        slen = np.ascontiguousarray(sequence_length.astype(np.int32, copy=False))
        featurized_x = torch.zeros((input_x.shape[0], self.init_rffs), dtype=torch.float32, device=self.device)
        ext.hipConv1dMaxpool(input_x, featurized_x, self.radem_diag1, self.chi_arr1, slen, self.conv_width)
        return featurized_x

    def transform_x(self, input_x, sequence_length=None):
        """kernel_baseclass.py:269-299 with l2_conv1d.py:150-185."""
        xin = self._as_device_f32(input_x).to(torch.float32, copy=True).contiguous()
        featurized_x = scale_input(self._first_layer(xin, sequence_length), self.hyperparams[1])
        xtrans = torch.zeros((featurized_x.shape[0], self.num_rffs), dtype=torch.float64, device=self.device)
        ext.hipRBFFeatureGen(featurized_x, xtrans, self.radem_diag, self.chi_arr, self.fit_intercept)
        if self.fit_intercept:
            xtrans[:, 0] = 1.
        return xtrans

    def kernel_specific_gradient(self, input_x, sequence_length=None):
        """l2_conv1d.py:189-222."""
        featurized_x = self._first_layer(input_x, sequence_length)
        output_x = torch.zeros((input_x.shape[0], self.num_rffs), dtype=torch.float64, device=self.device)
        dz_dsigma = torch.zeros((input_x.shape[0], self.num_rffs, 1), dtype=torch.float64, device=self.device)
        ext.hipRBFGrad(featurized_x, output_x, dz_dsigma, self.radem_diag, self.chi_arr,
                       float(self.hyperparams[1]), self.fit_intercept)
        return output_x, dz_dsigma

    # the resident float32 cache holds complete feature rows, as for the other sequence kernels
    def cache_ok(self):
        return self.num_freqs <= 16384

    def block_ok(self):
        return self.num_rffs % 4 == 0

    build_feature_cache = ConvSORFKernel.build_feature_cache
    ztz_matvec_cached = ConvSORFKernel.ztz_matvec_cached
    ztz_block_cached = ConvSORFKernel.ztz_block_cached
    cache_rows_to_features = ConvSORFKernel.cache_rows_to_features
    workspace_bytes = ConvSORFKernel.workspace_bytes


class MiniARDKernel(KernelBase):
    """kernels/ARD_kernels/mini_ard.py:16-287: an RBF kernel with one inverse lengthscale per group of
    input features (groups delimited by ``split_points``).  hyperparams = [lambda, sigma_1 .. sigma_G].
    Features: input scaled per feature, then the SORF operator; gradient: dense precomputed weights
    (three FHT rounds on the identity, built on the device with hipFastHadamardTransform2D in float64)
    through hipMiniARDGrad."""

    def __init__(self, xdim, num_rffs, random_seed=123, device="cuda", double_precision=False,
                 kernel_spec_parms=None):
        kernel_spec_parms = kernel_spec_parms or {}
        super().__init__(num_rffs, xdim, kernel_spec_parms, device)
        self.random_seed = random_seed
        self.double_precision = double_precision
        if len(self._xdim) != 2:
            raise ValueError("The dimensionality of the input is inappropriate for "
                             "the kernel you have selected.")
        if "split_points" not in kernel_spec_parms:
            raise ValueError("For the MiniARD kernel, 'kernel_specific_params' "
                             "must contain a list called 'split_points'.")
        if not isinstance(kernel_spec_parms["split_points"], list):
            raise ValueError("For the MiniARD kernel, 'split_points' must be a list.")
        self.kernel_choice = "MiniARD"
        self.split_pts = np.sort([0] + kernel_spec_parms["split_points"] + [xdim[1]])
        self._check_split_points(xdim)
        self.hyperparams = np.ones((self.split_pts.shape[0]))
        self.padded_dims = padded_dims(xdim[-1])
        radem_array = np.asarray([-1, 1], dtype=np.int8)
        rng = np.random.default_rng(random_seed)
        self.nblocks = ceil(self.num_freqs / self.padded_dims) if self.padded_dims < self.num_freqs else 1
        radem = rng.choice(radem_array, size=(3, 1, self.nblocks * self.padded_dims), replace=True)
        chi_arr = _chi.rvs(df=self.padded_dims, size=self.num_freqs, random_state=random_seed)
        if not double_precision:
            chi_arr = chi_arr.astype(np.float32)
        self._to_device(radem, chi_arr)
        self.precomputed_weights = None
        self._set_ard_arrays()

    def _check_split_points(self, xdim):
        """mini_ard.py:113-133."""
        if self.split_pts.shape[0] - 2 < 1:
            raise ValueError("There must be at least one split point to use MiniARD.")
        if self.split_pts[0] < 0:
            raise ValueError("The first split point must be > 0.")
        if self.split_pts[-1] > xdim[1]:
            raise ValueError("The last split point must be < shape[1] of the input data.")
        if np.diff(self.split_pts).min() == 0:
            raise ValueError("At least two of the split points supplied are identical.")

    def _set_ard_arrays(self):
        """mini_ard.py:158-168 (kernel_specific_set_hyperparams)."""
        full = np.zeros((self._xdim[-1]))
        key = np.zeros((self._xdim[-1]), dtype=np.int32)
        for i in range(1, self.split_pts.shape[0]):
            full[self.split_pts[i - 1]:self.split_pts[i]] = self.hyperparams[i]
            key[self.split_pts[i - 1]:self.split_pts[i]] = i - 1
        self.full_ard_weights = torch.from_numpy(full).to(self.device)
        self.ard_position_key = torch.from_numpy(key).to(self.device)

    def set_hyperparams(self, hyperparams, logspace=True):
        super().set_hyperparams(hyperparams, logspace)
        self._set_ard_arrays()

    supports_fused = False

    def fused_ok(self):
        return False

    def _typed(self, t):
        return t.to(torch.float64 if self.double_precision else torch.float32).contiguous()

    def transform_x(self, input_x, sequence_length=None):
        """kernel_baseclass.py:269-299 with mini_ard.py:171-194 (no sigma pre-multiplication: the
        per-feature weights carry the lengthscales)."""
        xin = self._typed(self._as_device(input_x))                 # the private typed copy (:274-288)
        xtrans = self._typed(xin.to(torch.float64) * self.full_ard_weights[None, :])
        output_x = torch.zeros((xtrans.shape[0], self.num_rffs), dtype=torch.float64, device=self.device)
        ext.hipRBFFeatureGen(xtrans, output_x, self.radem_diag, self._typed(self.chi_arr), self.fit_intercept)
        if self.fit_intercept:
            output_x[:, 0] = 1.
        return output_x

    def precompute_weights(self):
        """mini_ard.py:196-238, on the device in float64."""
        p = self.padded_dims
        norm_constant = 1.0 / (2.0 ** (np.log2(p) / 2.0))
        padded_chi = torch.zeros(self.nblocks * p, dtype=torch.float64, device=self.device)
        padded_chi[:self.chi_arr.shape[0]] = self.chi_arr.to(torch.float64)
        radem = self.radem_diag.to(torch.float64)
        blocks = []
        for i in range(self.nblocks):
            ident = torch.eye(p, dtype=torch.float64, device=self.device)
            lo, hi = i * p, (i + 1) * p
            for r in range(3):
                ident *= radem[r:r + 1, 0, lo:hi] * norm_constant
                ext.hipFastHadamardTransform2D(ident)
            ident *= padded_chi[lo:hi]
            blocks.append(ident.T[:, :self._xdim[-1]])
        self.precomputed_weights = self._typed(torch.cat(blocks)[:self.num_freqs, :])

    def kernel_specific_gradient(self, input_x, sequence_length=None):
        """mini_ard.py:240-275."""
        if self.precomputed_weights is None:
            self.precompute_weights()
        nl = self.split_pts.shape[0] - 1
        xtrans = torch.zeros((input_x.shape[0], self.num_rffs), dtype=torch.float64, device=self.device)
        dz_dsigma = torch.zeros((input_x.shape[0], self.num_rffs, nl), dtype=torch.float64, device=self.device)
        ext.hipMiniARDGrad(input_x, xtrans, self.precomputed_weights, self.ard_position_key,
                           self.full_ard_weights, dz_dsigma, self.fit_intercept)
        return xtrans, dz_dsigma

    def gradient_x(self, input_x, sequence_length=None):
        xin = self._typed(self._as_device(input_x))
        xtrans, xgrad = self.kernel_specific_gradient(xin, sequence_length)
        if self.fit_intercept:
            xtrans[:, 0] = 1.
            xgrad[:, 0, :] = 0.
        return xtrans, xgrad


_FIXED = ("RBF", "Matern", "Cauchy")
_CONV = ("Conv1dRBF", "Conv1dMatern", "Conv1dCauchy", "GraphRBF", "GraphMatern", "GraphCauchy")


def make_kernel(kernel_choice, xdim, num_rffs, random_seed=123, device="cuda", kernel_spec_parms=None):
    """Counterpart of KERNEL_NAME_TO_CLASS (kernels/__init__.py:21-33) for the SORF kernels."""
    if kernel_choice in _FIXED:
        return SORFKernel(kernel_choice, xdim, num_rffs, random_seed, device, kernel_spec_parms)
    if kernel_choice in _CONV:
        return ConvSORFKernel(kernel_choice, xdim, num_rffs, random_seed, device, kernel_spec_parms)
    if kernel_choice == "MiniARD":
        return MiniARDKernel(xdim, num_rffs, random_seed, device, False, kernel_spec_parms)
    if kernel_choice == "Conv1dTwoLayer":
        return Conv1dTwoLayerKernel(xdim, num_rffs, random_seed, device, kernel_spec_parms)
    if kernel_choice == "Linear":
        return LinearKernel(xdim, num_rffs, random_seed, device, kernel_spec_parms)
    raise RuntimeError(f"kernel '{kernel_choice}' is outside the hot path this package implements "
                       f"(supported: {_FIXED + _CONV + ('MiniARD', 'Conv1dTwoLayer', 'Linear')})")


class SRHTCompressor:
    """srht_compressor.py:37-97 -- double precision, as the preconditioner uses it."""

    def __init__(self, compression_size, input_size, device="cuda", random_seed=123):
        if compression_size >= input_size or compression_size <= 1:
            raise RuntimeError("The compression size should be < the number of rffs and > 1.")
        self.compression_size, self.input_size = compression_size, input_size
        self.padded_dims = padded_dims(input_size)
        radem_array = np.asarray([-1, 1], dtype=np.int8)
        rng = np.random.default_rng(random_seed)
        radem = rng.choice(radem_array, size=(self.padded_dims), replace=True)
        col_sampler = rng.permutation(self.padded_dims)
        self.device = device
        self.radem = torch.from_numpy(radem).to(device)
        self.col_sampler = torch.from_numpy(col_sampler).to(device)
        self.truncated_sampler = self.col_sampler[:compression_size].contiguous()
        self._zty_ws = None

    def fused_ok(self, features):
        return (features.is_cuda and features.dtype == torch.float64 and features.is_contiguous()
                and ext.srht_sample_ok(self.padded_dims, torch.float64))

    def transform_x_zty(self, features, ydata, zty_out, out=None):
        """The compressed chunk AND ``features.T @ ydata`` (into zty_out) from one read of the chunk
        (rand_nys_constructors.py:113-117)."""
        if self.fused_ok(features):
            if out is None:
                out = torch.empty((features.shape[0], self.compression_size), dtype=torch.float64,
                                  device=features.device)
            if self._zty_ws is None:
                self._zty_ws = torch.empty(ext.srht_sample_workspace_bytes(self.input_size), dtype=torch.uint8,
                                           device=features.device)
            ext.hipSRHTSample(features, self.radem, self.truncated_sampler, out, self.compression_size,
                              ydata, zty_out, self._zty_ws)
            return out[:, :self.compression_size]
        torch.matmul(features.T, ydata, out=zty_out)
        return self.transform_x(features, out=out)

    def transform_x(self, features, no_compression=False, out=None):
        """srht_compressor.py:87-97.  ``out``: optional preallocated float64 [n, >= compression_size] array
        whose leading columns receive the result (the fused pad + SRHT + gather operator writes into it)."""
        if features.dim() != 2 or features.shape[1] != self.input_size:
            raise RuntimeError("Input with unexpected size passed to a compressor module.")
        if not no_compression and self.fused_ok(features):
            if out is None:
                out = torch.empty((features.shape[0], self.compression_size), dtype=torch.float64,
                                  device=features.device)
            ext.hipSRHTSample(features, self.radem, self.truncated_sampler, out, self.compression_size)
            return out[:, :self.compression_size]
        if features.shape[1] < self.padded_dims:
            xfeatures = torch.zeros((features.shape[0], self.padded_dims), dtype=torch.float64,
                                    device=self.device)
            xfeatures[:, :features.shape[1]] = features
        else:
            xfeatures = features.to(torch.float64, copy=True).contiguous()
        ext.hipSRHT(xfeatures, self.radem)
        if no_compression:
            return xfeatures[:, self.col_sampler]
        if out is not None:
            out[:, :self.compression_size] = xfeatures[:, self.truncated_sampler]
            return out[:, :self.compression_size]
        return xfeatures[:, self.truncated_sampler]
