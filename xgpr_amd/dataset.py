"""Device-resident, row-sharded counterpart of the reference's OnlineDataset.

Mirrors data_handling/online_data_handling.py:54-94 (chunk generators; y standardised with
the stored training mean / std, :66-68) and data_handling/dataset_builder.py:14-178
(``build_regression_dataset``: trainy_mean = y.mean(), trainy_std = y.std()).  Differences,
both deliberate: the shard lives in HBM as float32 (the reference re-casts and re-uploads
every chunk on every CG iteration, kernels/kernel_baseclass.py:274-288), and with more than
one rank every rank holds a contiguous row range while the y statistics and
``get_ndatapoints()`` are global.
"""
import numpy as np
import torch

from .dist import SINGLE


class DeviceDataset:
    def __init__(self, xdata, ydata, sequence_lengths=None, chunk_size=2000,
                 trainy_mean=0.0, trainy_std=1.0, ndatapoints=None, device="cuda", comm=SINGLE, max_class=None):
        self.device = device
        self.comm = comm
        self._xdata = xdata
        self._ydata = ydata          # float64, raw (un-normalised), on the device
        self._sequence_lengths = sequence_lengths   # host int32 numpy array or None
        self._chunk_size = int(chunk_size)
        self._trainy_mean, self._trainy_std = float(trainy_mean), float(trainy_std)
        self._ndatapoints = int(ndatapoints if ndatapoints is not None else xdata.shape[0])
        self._max_class = max_class  # None for regression; the largest class label for classification
        self._scaled = {}

    def get_n_classes(self):
        """data_handling_baseclass.py:62-67."""
        return None if self._max_class is None else self._max_class + 1

    # ---- the reference's accessors (data_handling_baseclass.py)
    def get_ndatapoints(self):
        return self._ndatapoints

    def get_local_ndatapoints(self):
        return self._xdata.shape[0]

    def get_chunk_size(self):
        return self._chunk_size

    def get_xdim(self):
        return (self._ndatapoints,) + tuple(self._xdata.shape[1:])

    def get_ymean(self):
        return self._trainy_mean

    def get_ystd(self):
        return self._trainy_std

    def normalized_y(self):
        """(y - mean) / std as float64, the chunk-wise transform of
        online_data_handling.py:66-68 applied to the whole shard."""
        y = self._ydata.to(torch.float64).clone()
        y -= self._trainy_mean
        y /= self._trainy_std
        return y

    def get_chunked_data(self):
        n = self._xdata.shape[0]
        for i in range(0, n, self._chunk_size):
            j = min(i + self._chunk_size, n)
            if self._max_class is None:
                ychunk = self._ydata[i:j].to(torch.float64).clone()
                ychunk -= self._trainy_mean
                ychunk /= self._trainy_std
            else:                                  # class labels go through unchanged
                ychunk = self._ydata[i:j]
            lchunk = None if self._sequence_lengths is None else self._sequence_lengths[i:j]
            yield self._xdata[i:j, ...], ychunk, lchunk

    def get_chunked_x_data(self):
        n = self._xdata.shape[0]
        for i in range(0, n, self._chunk_size):
            j = min(i + self._chunk_size, n)
            lchunk = None if self._sequence_lengths is None else self._sequence_lengths[i:j]
            yield self._xdata[i:j, ...], lchunk

    def get_xdata(self):
        """The whole resident shard of x (device tensor)."""
        return self._xdata

    def get_sequence_lengths(self):
        """Host int32 sequence lengths of the shard, or None."""
        return self._sequence_lengths

    def feature_cache(self, kernel):
        """float32 feature cache of the whole shard for ``kernel`` at its current sigma (rebuilt when
        the kernel object or sigma changes)."""
        key = (id(kernel), float(kernel.hyperparams[1]))
        if getattr(self, "_zcache_key", None) != key:
            self._zcache = None
            self._zcache = kernel.build_feature_cache(self)
            self._zcache_key = key
        return self._zcache

    def get_chunked_features(self, kernel, with_y=False, from_cache=False):
        """Chunks of the float64 feature matrix (``kernel.transform_x`` of each x chunk), optionally with the
        standardised y chunk.  ``from_cache``: widen rows of the resident float32 feature cache instead of
        regenerating them -- for the convolution kernels, whose features cost K k-mers x SORF per sequence,
        this lets the preconditioner passes and the CG solve share one generation pass."""
        if not from_cache:
            for xin, yin, ldata in self.get_chunked_data():
                z = kernel.transform_x(xin, ldata)
                yield (z, yin) if with_y else z
            return
        # rows of the cache need no per-chunk generation, so they are served in chunks large enough for the dense
        # accumulations that consume them to run at full rate (sequence datasets use small chunks: 1024 sequences)
        zc = self.feature_cache(kernel)
        step = max(self._chunk_size, 8192)
        yall = None
        if with_y:
            yall = self._ydata if self._max_class is not None else self.normalized_y()
        for lo in range(0, zc.shape[0], step):
            z = kernel.cache_rows_to_features(zc[lo:lo + step])
            yield (z, yall[lo:lo + step]) if with_y else z

    def feature_cache_bytes(self, kernel):
        return self._xdata.shape[0] * kernel.get_num_rffs() * 4

    def scaled_x(self, sigma):
        """The whole shard pre-multiplied by sigma (what ``transform_x`` does to each chunk
        copy, sorf_kernel_baseclass.py:117), cached per sigma for the fused kernels."""
        from .kernels import scale_input, padded_dims
        key = float(sigma)
        if key not in self._scaled:
            xs = scale_input(self._xdata, key)
            # rows a multiple of four floats (zero columns appended, which is what the transform's own zero padding up
            # to the next power of two would have put there): 16-byte-aligned rows are what the three-wave fused matvec
            # fetches by LDS-DMA, so every input width gets that kernel
            if xs.dim() == 2 and xs.shape[1] % 4 != 0 and (xs.shape[1] + 3) // 4 * 4 <= padded_dims(xs.shape[1]):
                xp = torch.zeros((xs.shape[0], (xs.shape[1] + 3) // 4 * 4), dtype=xs.dtype, device=xs.device)
                xp[:, :xs.shape[1]] = xs
                xs = xp
            self._scaled = {key: xs}
        return self._scaled[key]


def build_regression_dataset(xdata, ydata, sequence_lengths=None, chunk_size=2000, device="cuda",
                             comm=SINGLE, already_sharded=False):
    """dataset_builder.py:14-178 for in-memory arrays.  ``xdata`` / ``ydata`` are numpy arrays
    or tensors; with ``comm.world_size > 1`` each rank keeps rows ``comm.shard_bounds(N)`` unless
    ``already_sharded`` (then the arrays passed are this rank's rows)."""
    xt = torch.from_numpy(np.ascontiguousarray(xdata)) if isinstance(xdata, np.ndarray) else xdata
    yt = torch.from_numpy(np.ascontiguousarray(ydata)) if isinstance(ydata, np.ndarray) else ydata
    if xt.shape[0] != yt.shape[0]:
        raise RuntimeError("Different number of datapoints in x and y.")
    sl = sequence_lengths
    if sl is not None:
        if xt.dim() != 3:
            raise RuntimeError("sequence_lengths supplied for a 2d array.")
        sl = np.ascontiguousarray(np.asarray(sl).astype(np.int32))
        if sl.max() > xt.shape[1] or sl.min() < 1:
            raise RuntimeError("sequence lengths out of range.")
    if comm.world_size > 1 and not already_sharded:
        lo, hi = comm.shard_bounds(xt.shape[0])
        xt, yt = xt[lo:hi], yt[lo:hi]
        if sl is not None:
            sl = sl[lo:hi]
    xt = xt.to(device=device, dtype=torch.float32).contiguous()
    yt = yt.to(device=device, dtype=torch.float64).contiguous()
    # global mean / population std (numpy's .std()), two passes in float64
    stats = torch.stack([yt.sum(), torch.tensor(float(yt.shape[0]), dtype=torch.float64, device=device)])
    comm.all_reduce_(stats)
    n_global = int(round(stats[1].item()))
    mean = stats[0] / stats[1]
    ssq = ((yt - mean) ** 2).sum().reshape(1)
    comm.all_reduce_(ssq)
    std = torch.sqrt(ssq[0] / stats[1])
    return DeviceDataset(xt, yt, sl, chunk_size, mean.item(), std.item(), n_global, device, comm)


def build_classification_dataset(xdata, ydata, sequence_lengths=None, chunk_size=2000, device="cuda",
                                 comm=SINGLE, already_sharded=False):
    """dataset_builder.py:68-117, :150-190 for in-memory arrays: integer labels in [0, max_class] with a
    zero category, no y normalisation.  Sharding as in ``build_regression_dataset``; the class count is
    global."""
    xt = torch.from_numpy(np.ascontiguousarray(xdata)) if isinstance(xdata, np.ndarray) else xdata
    yt = torch.from_numpy(np.ascontiguousarray(ydata)) if isinstance(ydata, np.ndarray) else ydata
    if yt.dim() != 1:
        raise RuntimeError("Y must be a 1d numpy array.")
    if yt.is_floating_point() or yt.dtype == torch.bool:
        raise RuntimeError("For classification, ydata must be an array of integers.")
    if xt.shape[0] != yt.shape[0]:
        raise RuntimeError("Different number of datapoints in x and y.")
    sl = sequence_lengths
    if sl is not None:
        if xt.dim() != 3:
            raise RuntimeError("sequence_lengths supplied for a 2d array.")
        sl = np.ascontiguousarray(np.asarray(sl).astype(np.int32))
        if sl.max() > xt.shape[1] or sl.min() < 1:
            raise RuntimeError("sequence lengths out of range.")
    if comm.world_size > 1 and not already_sharded:
        lo, hi = comm.shard_bounds(xt.shape[0])
        xt, yt = xt[lo:hi], yt[lo:hi]
        if sl is not None:
            sl = sl[lo:hi]
    xt = xt.to(device=device, dtype=torch.float32).contiguous()
    yt = yt.to(device=device, dtype=torch.int64).contiguous()
    # [max label, -min label, local row count]: max-reduce the first two, sum the third
    ext = torch.stack([yt.max(), -yt.min()]).to(torch.float64)
    comm.all_reduce_max_(ext)
    count = torch.tensor([float(yt.shape[0])], dtype=torch.float64, device=device)
    comm.all_reduce_(count)
    if int(-ext[1].item()) != 0:
        raise RuntimeError("For classification, there must be a zero category.")
    return DeviceDataset(xt, yt, sl, chunk_size, 0.0, 1.0, int(round(count.item())), device, comm,
                         max_class=int(ext[0].item()))


# ---------------------------------------------------------------------------------------------------
# On-disk datasets (data_handling/dataset_builder.py:193-340, offline_data_handling.py:73-108): lists of
# .npy chunk files.  The reference re-reads every file on every pass; here the files of this rank (whole
# files, round-robin over ranks) are streamed ONCE into the HBM-resident shard -- a loader thread reads
# file i+1 from disk into pinned host memory while file i is copied to the device on a side stream -- and
# everything downstream is the in-memory dataset.
# ---------------------------------------------------------------------------------------------------
def _check_offline_file(xfile, yfile, lfile, ndim, width, chunk_size):
    x = np.load(xfile)
    y = np.load(yfile)
    if x.shape[0] == 0:
        raise RuntimeError(f"File {xfile} has no datapoints.")
    if np.isnan(x).any():
        raise RuntimeError(f"One or more elements in file {xfile} is nan.")
    if np.max(x) > 1e15 or np.min(x) < -1e15:
        raise RuntimeError(f"One or more values in {xfile} is > 1e15 or < -1e15. Please check for inf values "
                           "and / or rescale your data.")
    if x.ndim != ndim:
        raise RuntimeError(f"File {xfile} is not a {ndim}d array, unlike some other arrays in xlist.")
    if tuple(x.shape[1:]) != tuple(width):
        raise RuntimeError("All x arrays must have the same dimensionality.")
    if x.shape[0] != y.shape[0]:
        raise RuntimeError(f"File {xfile} has a different number of datapoints than file {yfile}.")
    if x.shape[0] > chunk_size:
        raise RuntimeError(f"Xfile {xfile} has more datapoints than allowed based on specified chunk_size.")
    if y.ndim > 1:
        raise RuntimeError(f"The y file {yfile} is not a 1d array.")
    sl = None
    if lfile is not None:
        sl = np.load(lfile)
        if sl.shape[0] != x.shape[0] or sl.max() > x.shape[1] or sl.min() < 1:
            raise RuntimeError(f"The sequence lengths in {lfile} do not match {xfile}.")
    return x, y, sl


def build_offline_np_dataset(xlist, ylist, sequence_lengths=None, chunk_size=2000, device="cuda", comm=SINGLE,
                             task_type="regression", prefetch_depth=2):
    """dataset_builder.py:193-340 for lists of .npy files (one chunk per file, each <= chunk_size rows)."""
    import queue
    import threading
    if not isinstance(xlist, list) or not isinstance(ylist, list):
        raise RuntimeError("Both xlist and ylist should be lists.")
    if len(xlist) == 0:
        raise RuntimeError("At least one datafile must be supplied.")
    if len(xlist) != len(ylist):
        raise RuntimeError("xlist and ylist must have the same length.")
    if sequence_lengths is not None and len(sequence_lengths) != len(ylist):
        raise RuntimeError("sequence_lengths must either be None or have the same length as ylist.")
    lfiles = sequence_lengths if sequence_lengths is not None else [None] * len(ylist)
    first = np.load(xlist[0], mmap_mode="r")
    ndim, width = first.ndim, first.shape[1:]
    if ndim not in (2, 3):
        raise RuntimeError("Arrays should be either 2d or 3d.")
    mine = list(range(comm.rank, len(xlist), comm.world_size))      # whole files, round-robin over ranks
    rows = [np.load(xlist[i], mmap_mode="r").shape[0] for i in mine]
    n_local = int(sum(rows))
    is_cuda = torch.device(device).type == "cuda"
    xt = torch.empty((n_local,) + tuple(width), dtype=torch.float32, device=device)
    ydtype = torch.float64 if task_type == "regression" else torch.int64
    yt = torch.empty(n_local, dtype=ydtype, device=device)
    sl_all = np.empty(n_local, dtype=np.int32) if sequence_lengths is not None else None

    q = queue.Queue(maxsize=max(1, prefetch_depth))

    def loader():
        try:
            for i in mine:
                x, y, sl = _check_offline_file(xlist[i], ylist[i], lfiles[i], ndim, width, chunk_size)
                if task_type == "classification" and not np.issubdtype(y.dtype, np.integer):
                    raise RuntimeError("For classification, there must be a zero category, and all yfiles "
                                       "must be integers.")
                xh = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
                yh = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64 if task_type == "regression" else np.int64))
                if is_cuda:
                    xh, yh = xh.pin_memory(), yh.pin_memory()
                q.put((xh, yh, sl))
            q.put(None)
        except Exception as err:      # surface loader errors in the caller's thread
            q.put(err)

    th = threading.Thread(target=loader, daemon=True)
    th.start()
    copy_stream = torch.cuda.Stream(device=device) if is_cuda else None
    row = 0
    while True:
        item = q.get()
        if item is None:
            break
        if isinstance(item, Exception):
            raise item
        xh, yh, sl = item
        j = row + xh.shape[0]
        if is_cuda:
            with torch.cuda.stream(copy_stream):
                xt[row:j].copy_(xh, non_blocking=True)
                yt[row:j].copy_(yh, non_blocking=True)
            copy_stream.synchronize()         # the pinned buffers are released when xh / yh go out of scope
        else:
            xt[row:j] = xh
            yt[row:j] = yh
        if sl_all is not None:
            sl_all[row:j] = sl.astype(np.int32)
        row = j
    th.join()
    if task_type == "classification":
        return build_classification_dataset(xt, yt, sl_all, chunk_size, device, comm, already_sharded=True)
    return build_regression_dataset(xt, yt, sl_all, chunk_size, device, comm, already_sharded=True)
