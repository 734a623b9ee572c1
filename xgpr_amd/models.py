"""The reference's two model classes as thin state holders over this package's functions, so that a script
written against xGPR reads the same here:

    xGPRegression      <-> xgp_regression.py (fit / predict / exact_nmll / exact_nmll_gradient /
                            approximate_nmll / tune_hyperparams, build_preconditioner, set_hyperparams)
    xGPClassification  <-> xgp_classification.py (fit / predict)

Differences, all deliberate: datasets are the HBM-resident ones of ``xgpr_amd.dataset`` (built with
``build_regression_dataset`` / ``build_classification_dataset`` / ``build_offline_np_dataset``), the device is the
HIP device (there is no CPU mode), and inputs / outputs at the API boundary are numpy arrays as in the reference.
Everything numerical lives in kernels.py, cg.py, preconditioner.py, exact.py, nmll.py, classification.py, tuning.py.
"""
import numpy as np
import torch

from . import nmll as _nmll
from .cg import cg_fit_lib_internal
from .classification import fit_classifier, predict_proba
from .exact import calc_weights_exact, calc_variance_exact
from .kernels import make_kernel
from .preconditioner import RandNysPreconditioner, autoselect_preconditioner
from .crude_tuning import tune_hyperparams_crude as _tune_crude
from .tuning import tune_hyperparams as _tune

MAX_VARIANCE_RFFS = 4096            # constants.py:2
MAX_CLOSED_FORM_RFFS = 8192         # constants.py:3
DEFAULT_KERNEL_SPEC_PARMS = {"matern_nu": 5 / 2, "intercept": True, "averaging": "none"}      # constants.py:7-8


class _ModelBase:
    """model_baseclass.py:68-125 (constructor arguments), :171-222 (hyperparameters), :225-260
    (build_preconditioner)."""
    is_regression = True

    def __init__(self, num_rffs=256, variance_rffs=16, kernel_choice="RBF", device="cuda", kernel_settings=None,
                 verbose=True, random_seed=123):
        if kernel_settings is not None and not isinstance(kernel_settings, dict):
            raise RuntimeError("kernel_settings must be a dict.")
        if variance_rffs > MAX_VARIANCE_RFFS:
            raise RuntimeError("Currently to keep computational expense at acceptable levels variance rffs is "
                               f"capped at {MAX_VARIANCE_RFFS}.")
        self.num_rffs, self.variance_rffs = num_rffs, variance_rffs
        self.kernel_choice, self.device = kernel_choice, device
        self.kernel_spec_parms = dict(DEFAULT_KERNEL_SPEC_PARMS if kernel_settings is None else kernel_settings)
        self.verbose, self.random_seed = verbose, random_seed
        self.kernel = None
        self.weights = self.var = self.gamma = None
        self.trainy_mean, self.trainy_std = 0.0, 1.0

    def _initialize_kernel(self, dataset):
        if self.kernel is None:
            self.kernel = make_kernel(self.kernel_choice, dataset.get_xdim(), self.num_rffs, self.random_seed,
                                      self.device, self.kernel_spec_parms)

    def set_hyperparams(self, hyperparams=None, dataset=None):
        """model_baseclass.py:171-213: log-space hyperparameters; the kernel is created from the dataset's
        dimensions on first use."""
        if self.kernel is None:
            if dataset is None:
                raise RuntimeError("A dataset is required if the kernel has not already been initialized.")
            self._initialize_kernel(dataset)
        if hyperparams is not None:
            if not isinstance(hyperparams, np.ndarray) or hyperparams.shape != self.kernel.get_hyperparams().shape:
                raise RuntimeError("The hyperparameters must be a numpy array of the kernel's hyperparameter shape.")
            self.kernel.set_hyperparams(hyperparams, logspace=True)
        self.weights = self.var = self.gamma = None

    def get_hyperparams(self):
        return None if self.kernel is None else self.kernel.get_hyperparams()

    def build_preconditioner(self, dataset, max_rank=512, method="srht"):
        """-> (preconditioner, achieved_ratio)"""
        self._initialize_kernel(dataset)
        if max_rank < 1:
            raise RuntimeError("Invalid value for max_rank.")
        if max_rank >= self.kernel.get_num_rffs():
            raise RuntimeError("Max rank should be < the number of rffs.")
        pre = RandNysPreconditioner(self.kernel, dataset, max_rank, self.verbose, self.random_seed, method,
                                    is_regression=self.is_regression)
        return pre, pre.achieved_ratio

    def _to_device(self, arr):
        return arr if isinstance(arr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(arr))


class xGPRegression(_ModelBase):
    def fit(self, dataset, preconditioner=None, tol=1e-6, max_iter=500, mode="cg", suppress_var=False,
            max_rank=3000, min_rank=512, autoselect_target_ratio=30., always_use_srht2=False, run_diagnostics=False,
            cache_features="auto"):
        """xgp_regression.py:381-493."""
        self._initialize_kernel(dataset)
        self.trainy_mean, self.trainy_std = dataset.get_ymean(), dataset.get_ystd()
        self.weights = self.var = None
        if mode == "exact":
            if self.kernel.get_num_rffs() > MAX_CLOSED_FORM_RFFS:
                raise RuntimeError(f"You specified 'exact' fitting, but the number of rffs is > {MAX_CLOSED_FORM_RFFS}.")
            self.weights, n_iter, losses = calc_weights_exact(dataset, self.kernel)
        elif mode == "cg":
            if preconditioner is None:
                preconditioner, _, _ = autoselect_preconditioner(
                    self.kernel, dataset, min_rank, max_rank, 512, always_use_srht2, autoselect_target_ratio,
                    self.random_seed, True, self.verbose)
            self.weights, n_iter, losses = cg_fit_lib_internal(self.kernel, dataset, tol, max_iter, preconditioner,
                                                               self.verbose, cache_features=cache_features)
        else:
            raise RuntimeError("Unrecognized fitting mode supplied. Must provide one of 'cg', 'exact'.")
        if not suppress_var:
            nvar = min(self.variance_rffs, self.kernel.get_num_rffs())
            self.var = calc_variance_exact(self.kernel, dataset, nvar)
        if run_diagnostics:
            return n_iter, losses

    def predict(self, input_x, sequence_lengths=None, get_var=False, chunk_size=2000):
        """xgp_regression.py:77-148 -> numpy predictions (and variances)."""
        if self.weights is None:
            raise RuntimeError("Model has not yet been successfully fitted.")
        if get_var and self.var is None:
            raise RuntimeError("Variance was requested but suppress_var was selected when fitting.")
        lambda_ = float(self.kernel.get_lambda())
        preds, var = [], []
        for i in range(0, input_x.shape[0], chunk_size):
            sl = None if sequence_lengths is None else sequence_lengths[i:i + chunk_size]
            xfeatures = self.kernel.transform_x(input_x[i:i + chunk_size], sl)
            preds.append((xfeatures * self.weights[None, :]).sum(dim=1))
            if get_var:
                xv = xfeatures[:, :self.var.shape[0]]
                pred_var = (self.var @ xv.T).T
                var.append(lambda_ ** 2 + lambda_ ** 2 * (xv * pred_var).sum(dim=1))
        preds = torch.cat(preds).cpu().numpy() * self.trainy_std + self.trainy_mean
        if not get_var:
            return preds
        var = torch.cat(var).cpu().numpy()
        var[var < 0] = 0
        return preds, var * self.trainy_std ** 2

    def exact_nmll(self, hyperparams, dataset):
        self.set_hyperparams(hyperparams, dataset)
        return _nmll.exact_nmll(self.kernel, dataset)

    def exact_nmll_gradient(self, hyperparams, dataset):
        self.set_hyperparams(hyperparams, dataset)
        return _nmll.exact_nmll_gradient(self.kernel, dataset)

    def approximate_nmll(self, hyperparams, dataset, manual_settings=None):
        self.set_hyperparams(hyperparams, dataset)
        return _nmll.approximate_nmll(self.kernel, dataset, None, manual_settings, self.random_seed)

    def tune_hyperparams(self, dataset, bounds=None, max_iter=50, tuning_method="Powell", starting_hyperparams=None,
                         tol=1e-2, n_restarts=1, nmll_method="exact", manual_settings=None):
        self._initialize_kernel(dataset)
        self.weights = self.var = None
        return _tune(self.kernel, dataset, bounds, max_iter, tuning_method, starting_hyperparams, tol, n_restarts,
                     nmll_method, manual_settings, self.random_seed, self.verbose)


    def tune_hyperparams_crude(self, dataset, bounds=None, random_seed=123, max_bayes_iter=30, subsample=1):
        """xgp_regression.py:497-561."""
        self._initialize_kernel(dataset)
        self.weights = self.var = None
        return _tune_crude(self.kernel, dataset, bounds, random_seed, max_bayes_iter, subsample, self.verbose)


class xGPClassification(_ModelBase):
    is_regression = False

    def __init__(self, num_rffs=256, kernel_choice="RBF", device="cuda", kernel_settings=None, verbose=True,
                 random_seed=123):
        super().__init__(num_rffs, 0, kernel_choice, device, kernel_settings, verbose, random_seed)

    def fit(self, dataset, preconditioner=None, tol=1e-3, max_iter=500, max_rank=3000, min_rank=512,
            autoselect_target_ratio=30., always_use_srht2=False, run_diagnostics=False, cache_features="auto"):
        """xgp_classification.py:111-200."""
        self._initialize_kernel(dataset)
        if preconditioner is None:
            preconditioner, _, _ = autoselect_preconditioner(
                self.kernel, dataset, min_rank, max_rank, 512, always_use_srht2, autoselect_target_ratio,
                self.random_seed, False, self.verbose)
        self.weights, self.gamma, n_iter, losses = fit_classifier(self.kernel, dataset, preconditioner, tol, max_iter,
                                                                  self.verbose, cache_features)
        if run_diagnostics:
            return n_iter, losses

    def predict(self, input_x, sequence_lengths=None, chunk_size=2000):
        """xgp_classification.py:59-109 -> numpy [N, classes] probabilities."""
        if self.gamma is None:
            raise RuntimeError("Model has not been fitted yet.")
        return predict_proba(self.kernel, self.weights, self.gamma, self._to_device(input_x), sequence_lengths,
                             chunk_size).cpu().numpy()


class KernelFGen:
    """kernel_fgen.py / auxiliary_baseclass.py:27-92: random features of a chosen kernel for use outside a model
    (kernel k-means, PCA): no intercept column, kernel-specific hyperparameters supplied by the caller."""

    def __init__(self, num_rffs, hyperparams, num_features, kernel_choice="RBF", device="cuda", kernel_settings=None,
                 random_seed=123, verbose=True):
        settings = dict(DEFAULT_KERNEL_SPEC_PARMS if kernel_settings is None else kernel_settings)
        settings["intercept"] = False
        three_d = kernel_choice.startswith(("Conv1d", "Graph"))
        xdim = (1, settings.get("conv_width", 10), num_features) if three_d else (1, num_features)
        self.kernel = make_kernel(kernel_choice, xdim, num_rffs, random_seed, device, settings)
        self.device, self.verbose = device, verbose
        full = self.kernel.get_hyperparams()
        if full.shape[0] > 1:
            full[1:] = hyperparams
        self.kernel.set_hyperparams(full, logspace=True)

    def predict(self, input_x, sequence_lengths=None, chunk_size=2000):
        """-> numpy [N, num_rffs]"""
        preds = []
        for i in range(0, input_x.shape[0], chunk_size):
            sl = None if sequence_lengths is None else sequence_lengths[i:i + chunk_size]
            preds.append(self.kernel.transform_x(input_x[i:i + chunk_size], sl))
        return torch.cat(preds).cpu().numpy()


class FastConv1d:
    """static_layers/fast_conv.py with kernels/convolution_kernels/conv_feature_extractor.py:37-108: the static
    convolution + global max-pool feature extractor (hipConv1dMaxpool) for sequences."""

    def __init__(self, seq_width, device="cuda", random_seed=123, conv_width=9, num_features=512):
        from math import ceil
        from scipy.stats import chi as _chi
        from .kernels import padded_dims
        self.seq_width, self.num_features, self.conv_width, self.device = seq_width, num_features, conv_width, device
        rng = np.random.default_rng(random_seed)
        pdims = padded_dims(conv_width * seq_width)
        radem = rng.choice(np.asarray([-1, 1], dtype=np.int8), size=(3, 1, ceil(num_features / pdims) * pdims),
                           replace=True)
        chi_arr = _chi.rvs(df=pdims, size=num_features, random_state=random_seed).astype(np.float32)
        self.radem_diag = torch.from_numpy(np.ascontiguousarray(radem)).to(device)
        self.chi_arr = torch.from_numpy(chi_arr).to(device)

    def predict(self, x_array, sequence_lengths, chunk_size=2000):
        """-> numpy float32 [N, num_features]"""
        from . import xgpr_hip_rfgen_ext as ext
        if sequence_lengths.shape[0] != x_array.shape[0]:
            raise RuntimeError("The shape[0] of sequence_lengths must match the shape[0] of x_array.")
        if x_array.shape[2] != self.seq_width:
            raise ValueError("Unexpected number of features per timepoint / sequence element on this input.")
        feats = []
        for i in range(0, x_array.shape[0], chunk_size):
            xin = x_array[i:i + chunk_size]
            xin = torch.from_numpy(np.ascontiguousarray(xin)) if isinstance(xin, np.ndarray) else xin
            xin = xin.to(self.device, torch.float32).contiguous()
            out = torch.zeros((xin.shape[0], self.num_features), dtype=torch.float32, device=self.device)
            ext.hipConv1dMaxpool(xin, out, self.radem_diag, self.chi_arr,
                                 np.ascontiguousarray(np.asarray(sequence_lengths[i:i + chunk_size]).astype(np.int32)),
                                 self.conv_width)
            feats.append(out)
        return torch.cat(feats).cpu().numpy()
