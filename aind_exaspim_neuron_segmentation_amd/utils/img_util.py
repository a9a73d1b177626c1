"""
Host-side helpers of the prediction path that mirror the reference's
``utils/img_util.py`` (normalize: 504-533, get_patch_slices: 405-428,
add_padding: 362-379).

On the product path the voxel arithmetic of these functions (brightness clip,
percentile normalisation, clipping to [0, 1], reflect padding, float32 cast)
runs fused inside the HIP gather kernel (csrc/prepost.hip). What stays on the
host is integer bookkeeping: patch slices, and turning the device histogram
into the two percentile values exactly as ``numpy.percentile`` would.
"""

import numpy as np


def get_patch_slices(start, patch_shape, img_shape):
    """
    Computes slices for a 3D patch within an image, clipped to the image
    boundaries.

    Parameters
    ----------
    start : Tuple[int]
        Starting indices (z, y, x) of the patch.
    patch_shape : Tuple[int]
        Desired patch shape (depth, height, width).
    img_shape : Tuple[int]
        Shape of the image that the patch is contained within.

    Returns
    -------
    Tuple[slice]
        Slices to index the image: (slice_z, slice_y, slice_x).
    """
    return tuple(
        slice(s, min(s + ps, d)) for s, ps, d in zip(start, patch_shape, img_shape)
    )


def is_contained(voxel, shape, buffer=0):
    """
    Checks whether a voxel is within bounds of a given shape, considering a
    buffer (img_util.py:451-474): the test the reference's validation tiling
    applies to the centres of the zero-overlap patch grid
    (data_handling.py:402-413).

    Parameters
    ----------
    voxel : Tuple[int]
        Voxel coordinates to be checked.
    shape : Tuple[int]
        Shape of the image volume.
    buffer : int, optional
        Number of voxels to pad the bounds by. Default is 0.

    Returns
    -------
    bool
        True if voxel - buffer and voxel + buffer both lie inside [0, size) on
        every axis (two-sided on both, so a negative buffer is handled like the
        reference handles it).
    """
    return all(0 <= v + buffer < s and 0 <= v - buffer < s for v, s in zip(voxel, shape))


def reflect_index(j, n):
    """
    Source index inside a length-n axis for position j of its high-side
    'reflect' padding (what numpy.pad(mode="reflect") reads); the arithmetic
    rule the gather kernel implements.
    """
    if n == 1:
        return 0
    period = 2 * (n - 1)
    m = j % period
    return m if m < n else period - m


class OrderStatistics:
    """
    Exact order statistics of a volume from 65536-bin histograms.

    Parameters
    ----------
    hist : numpy.ndarray
        int64/uint64 counts of shape (65536,) (whole volume, all ranks).
    bin_to_value : Callable[[int], numpy.generic]
        Maps a bin index to the voxel value (as a numpy scalar of the voxel
        dtype after the brightness clip).
    """

    def __init__(self, hist, bin_to_value):
        self.hist = np.asarray(hist).astype(np.int64)
        self.cum = np.cumsum(self.hist)
        self.n = int(self.cum[-1]) if self.cum.size else 0
        self.bin_to_value = bin_to_value

    def bin_of_rank(self, k):
        """Bin holding the k-th smallest voxel (0-based), and rank within it."""
        b = int(np.searchsorted(self.cum, k, side="right"))
        below = int(self.cum[b - 1]) if b > 0 else 0
        return b, k - below

    def kth(self, k):
        """Value of the k-th smallest voxel (0-based)."""
        b, _ = self.bin_of_rank(k)
        return self.bin_to_value(b)


def percentiles_from_statistics(stats, percentiles, dtype):
    """
    Evaluates numpy.percentile(volume, percentiles) (method "linear") from
    order statistics, with the same ufunc calls in the same dtypes as numpy
    (numpy/lib/_function_base_impl.py: percentile, _quantile,
    the "linear" entry of _QuantileMethods, _get_indexes, _get_gamma,
    _lerp), so the float64 results are bit-identical.

    Parameters
    ----------
    stats : OrderStatistics
        Anything with attribute ``n`` and method ``kth(k)`` returning numpy
        scalars of the voxel dtype.
    percentiles : ArrayLike
        Percentiles in [0, 100].
    dtype : numpy.dtype
        Voxel dtype of the (clipped) volume.

    Returns
    -------
    numpy.ndarray
        One value per percentile.
    """
    dtype = np.dtype(dtype)
    n = stats.n
    if n == 0:
        raise ValueError("cannot take percentiles of an empty volume")
    q = np.true_divide(percentiles, dtype.type(100) if dtype.kind == "f" else 100)
    q = np.atleast_1d(np.asanyarray(q))
    if not (np.all(0 <= q) and np.all(q <= 1)):
        raise ValueError("Percentiles must be in the range [0, 100]")
    virtual = np.asanyarray((n - 1) * q)  # numpy's "linear" virtual index
    prev_f = np.asanyarray(np.floor(virtual))
    next_f = np.asanyarray(prev_f + 1)
    above = virtual >= n - 1
    prev_i = prev_f.copy()
    next_i = next_f.copy()
    prev_i[above] = n - 1
    next_i[above] = n - 1
    below = virtual < 0
    prev_i[below] = 0
    next_i[below] = 0
    previous = np.array([stats.kth(int(k)) for k in prev_i], dtype=dtype)
    nxt = np.array([stats.kth(int(k)) for k in next_i], dtype=dtype)
    gamma_prev = prev_f.copy()
    gamma_prev[above] = -1
    gamma_prev[below] = 0
    gamma = np.asanyarray(virtual - gamma_prev, dtype=virtual.dtype)
    diff = np.subtract(nxt, previous)
    result = np.asanyarray(np.add(previous, diff * gamma))
    np.subtract(nxt, diff * (1 - gamma), out=result, where=gamma >= 0.5,
                casting="unsafe", dtype=type(result.dtype))
    return result
