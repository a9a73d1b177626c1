"""
MI355X-native drop-in for the prediction section of the reference's
``inference.py`` (predict: 29-126, _predict_batch: 129-163,
_get_batch_inputs: 166-192, count_patches: 340-365, generate_patch_starts:
368-397, load_model: 400-424, to_tensor: 427-446).

Same function names, positional order, keyword names and defaults; extra
options are trailing keyword arguments. What differs is where the work runs:

* the volume is uploaded once and stays in HBM; brightness clip, percentile
  normalisation, patch extraction with reflect padding and the float32 cast
  happen in one HIP gather kernel per batch (exact float64 arithmetic, so the
  network sees bit-identical inputs);
* the two percentiles come from a device histogram (exact order statistics ->
  numpy's interpolation on the host);
* the U-Net runs on the hand-written gfx950 kernels behind ``UNet3D``;
* sigmoid, trimming, overlap-add and the final division happen on the device;
* the sliding window is z-major (inference.py:368-397), so an output slab is final
  one patch layer after the last patch touching it: finished slabs are divided,
  copied to pinned memory on a copy stream and moved into the result by a few host
  threads while the next layer computes (predict_streaming; predict() of a host
  array is that pipeline over the array).

There is no CPU fallback: a model on a CPU device raises.
"""

import itertools
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from aind_exaspim_neuron_segmentation_amd import _native
from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D
from aind_exaspim_neuron_segmentation_amd.utils import img_util

try:  # progress bar is cosmetic; the reference imports tqdm unconditionally
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    tqdm = None

# Batches in flight on separate HIP streams in predict() / predict_streaming(): measured on MI355X
# (1024^3, fp16, batch 16) three are 5 % faster than one -- one kernel's last, partly idle round of
# workgroups overlaps another batch's work -- and the result does not change by a bit (stitching
# stays in batch order on the caller's stream).
DEFAULT_STREAMS = 3
# Switches between bit-identical (PLAIN_GATHER) or superset (FULL_PATCHES) execution plans, for the tests that
# hold the plans to each other and for measurements; module attributes, not environment variables, so nothing
# outside the calling code can flip them.
PLAIN_GATHER = False    # True: gather float32 patches and let the engine pad them, instead of the prepared layout
FULL_PATCHES = False    # True: compute the margin predict() discards as well (exaspim_unet_forward untrimmed)

_VOX_CODES = {
    np.dtype(np.uint8): _native.VOX_U8,
    np.dtype(np.uint16): _native.VOX_U16,
    np.dtype(np.int16): _native.VOX_I16,
    np.dtype(np.float32): _native.VOX_F32,
    np.dtype(np.float64): _native.VOX_F64,
}
_TORCH_VOXELS = {np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.int16,
                 np.dtype(np.int16): torch.int16, np.dtype(np.float32): torch.float32,
                 np.dtype(np.float64): torch.float64}


# --- Model Predictions ---
def predict(
    img,
    model,
    affinity_mode=True,
    batch_size=16,
    brightness_clip=1000,
    normalization_percentiles=(1, 99.9),
    patch_shape=(96, 96, 96),
    overlap=(32, 32, 32),
    trim=8,
    verbose=True,
    *,
    return_device_tensor=False,
    n_streams=DEFAULT_STREAMS,
    out_dtype=np.float32,
):
    """
    Predicts affinities or foreground-background maps for a 3D image by
    splitting it into overlapping patches, batching the patches, and
    processing each batch with the model.

    Parameters
    ----------
    img : numpy.ndarray
        Input 3D image with shape (D, H, W), (1, D, H, W) or (1, 1, D, H, W).
        A torch tensor already on the model's device is accepted too.
    model : torch.nn.Module
        Model used for prediction (normally from "load_model"); it must live
        on a HIP device.
    affinity_mode : bool, optional
        If True, the model predicts affinities; if False, it predicts
        foreground-background. Default is True.
    batch_size : int, optional
        Number of patches to process in a batch. Default is 16.
    brightness_clip : float, optional
        Maximum brightness value for voxel intensities. Default is 1000.
    normalization_percentiles : Tuple[int], optional
        Lower and upper percentiles used for normalization. Default is
        (1, 99.9).
    patch_shape : Tuple[int], optional
        Shape of 3D patch expected by the model. Default is (96, 96, 96).
    overlap : Tuple[int], optional
        Shape of overlap between patches along each dimension. Default is
        (32, 32, 32).
    trim : int, optional
        Number of voxels to trim from the edges of each patch in the output.
        Default is 8.
    verbose : bool, optional
        Indication of whether to show a tqdm progress bar. Default is True.
    return_device_tensor : bool, optional
        Return the result as a torch tensor on the device instead of copying
        it to a numpy array (the whole accumulator then stays in HBM). Default
        is False: finished z-slabs are downloaded while later patch layers
        compute (see predict_streaming).
    n_streams : int, optional
        Batches in flight on separate HIP streams (see run_sliding_window);
        the result does not depend on it. Default is 3 (DEFAULT_STREAMS).
    out_dtype : numpy.dtype, optional
        numpy.float32 (default, the reference's) or numpy.float16: the
        finished result rounded to IEEE half on the device (round to nearest
        even, error at most 2.4e-4 on values in [0, 1]) before it leaves it --
        half the download and half the host memory. The consumer,
        affinities_to_segmentation, starts with astype(np.float32)
        (inference.py:223) and takes such an array as it is.

    Returns
    -------
    pred : numpy.ndarray
        Prediction generated by the given model applied to an image: float32
        (3, D, H, W), or (D, H, W) if "affinity_mode" is False.
    """
    out_dtype = _checked_out_dtype(out_dtype)
    device = next(model.parameters()).device
    if device.type != "cuda":
        raise RuntimeError(
            "predict (MI355X) has no CPU path: the model must be on a HIP device, "
            f"got {device}"
        )
    if not return_device_tensor and not isinstance(img, (DeviceVolume, torch.Tensor)):
        # host array in, host array out: slab pipeline with overlapped transfers
        return predict_streaming(
            img, model, affinity_mode=affinity_mode, batch_size=batch_size,
            brightness_clip=brightness_clip, normalization_percentiles=normalization_percentiles,
            patch_shape=patch_shape, overlap=overlap, trim=trim, verbose=verbose,
            n_streams=n_streams, out_dtype=out_dtype,
        )
    volume = DeviceVolume.from_array(img, device, clip=brightness_clip)
    plan = SlidingWindow(volume.shape, patch_shape, overlap, trim)
    n_channels = 3 if affinity_mode else 1
    with torch.cuda.device(device):
        mn, mx = volume_percentiles(volume, brightness_clip, normalization_percentiles)
        accum = run_sliding_window(
            volume, model, plan, n_channels, batch_size, brightness_clip, mn, mx,
            verbose=verbose, n_streams=n_streams,
        )
        stitch_finalize(accum, plan, volume.block)
        if out_dtype == np.float16:
            accum = export_half(accum)
    pred = accum if affinity_mode else accum[0]
    if return_device_tensor:
        return pred
    return pred.cpu().numpy()


def _checked_out_dtype(out_dtype):
    out_dtype = np.dtype(out_dtype)
    if out_dtype not in (np.dtype(np.float32), np.dtype(np.float16)):
        raise TypeError(f"out_dtype must be numpy.float32 or numpy.float16, got {out_dtype}")
    return out_dtype


def export_half(tensor, out=None):
    """
    Rounds a finalised float32 device tensor to IEEE half (round to nearest
    even, numpy's astype(float16)) with the library's export kernel.

    Parameters
    ----------
    tensor : torch.Tensor
        Contiguous float32 tensor on a HIP device.
    out : torch.Tensor, optional
        Contiguous float16 tensor with as many elements to write into.

    Returns
    -------
    torch.Tensor
        float16 tensor of the same shape.
    """
    if tensor.dtype != torch.float32 or not tensor.is_contiguous():
        raise ValueError("export_half needs a contiguous float32 tensor")
    if out is None:
        out = torch.empty(tensor.shape, dtype=torch.float16, device=tensor.device)
    elif out.dtype != torch.float16 or not out.is_contiguous() or out.numel() != tensor.numel():
        raise ValueError("export_half: out must be a contiguous float16 tensor of the same size")
    _native.check(
        _native.lib().exaspim_export_f16(tensor.data_ptr(), out.data_ptr(), tensor.numel(),
                                         _stream(tensor.device)),
        "exaspim_export_f16",
    )
    return out.view(tensor.shape)


def _predict_batch(img, model, starts, patch_shape, trim=8, *, clip=None, mn=0.0, mx=1.0):
    """
    Extracts a batch of 3D patches from a device-resident volume, runs them
    through the model and returns sigmoid predictions, trimmed if requested
    (inference.py:129-163 of the reference; there "img" is the normalised
    float64 volume, here normalisation is fused into the extraction).

    Parameters
    ----------
    img : DeviceVolume
        Volume in device memory.
    model : torch.nn.Module
        Model used for prediction.
    starts : List[Tuple[int]]
        Starting coordinates (z, y, x) of the patches.
    patch_shape : Tuple[int]
        Shape of 3D patch expected by the model.
    trim : int, optional
        Number of voxels trimmed from each side of the outputs. Default is 8.

    Returns
    -------
    torch.Tensor
        Device tensor (B, C, d, h, w) of predictions.
    """
    device = next(model.parameters()).device
    starts_dev = torch.tensor(list(starts), dtype=torch.int32, device=device).reshape(-1, 3)
    inputs = _get_batch_inputs(img, starts_dev, patch_shape, device, clip=clip, mn=mn, mx=mx)
    outputs = _model_probabilities(model, inputs, trim)
    if trim > 0:
        outputs = outputs[..., trim:-trim, trim:-trim, trim:-trim]
    return outputs


def _get_batch_inputs(img, starts, patch_shape, device, *, clip=None, mn=0.0, mx=1.0, out=None,
                      layout=_native.IN_F32):
    """
    Builds the (B, 1, *patch_shape) float32 network input for a batch of patch
    starts with the HIP gather kernel: brightness clip, normalisation, clip to
    [0, 1], in-volume slicing, high-side reflect padding and the float32 cast
    (inference.py:166-192, img_util.py:362-379, 405-428, 504-533).

    Parameters
    ----------
    img : DeviceVolume
        Volume in device memory.
    starts : torch.Tensor
        int32 device tensor (B, 3) of global patch starts.
    patch_shape : Tuple[int]
        Shape of the 3D patch expected by the model.
    device : torch.device
        Device of the result.
    layout : int, optional
        _native.IN_F32 (default: the reference's float32 batch) or the operand
        layout of a UNet3D's first convolution (UNet3D.input_layout(): the same
        values with a one-voxel zero border, split into 16-bit parts for the
        16-bit engines), which UNet3D.run_prepared takes.

    Returns
    -------
    torch.Tensor
        Float32 device tensor (B, 1, *patch_shape), or for a prepared layout a
        4-byte-per-voxel tensor (B, patch_shape[0] + 2, patch_shape[1] + 2,
        patch_shape[2] + 2).
    """
    n = int(starts.shape[0])
    border = 0 if layout == _native.IN_F32 else 2
    if out is None:
        shape = (n, 1) + tuple(patch_shape) if border == 0 else (n,) + tuple(int(p) + 2 for p in patch_shape)
        out = torch.empty(shape, dtype=torch.float32, device=device)
    has_clip = clip is not None
    denom = float(np.float64(mx) - np.float64(mn) + 1e-8)  # img_util.py:527
    # one launch takes n * (padded) patch depth <= 65535 grid rows: very large batches of
    # small patches go in pieces
    piece = max(1, 65535 // (int(patch_shape[0]) + border))
    for i in range(0, n, piece):
        m = min(piece, n - i)
        _native.check(
            _native.lib().exaspim_gather_patches_as(
                img.tensor.data_ptr(), img.vox_code, img.block, starts[i:i + m].data_ptr(), m,
                _native.int3(patch_shape), float(clip) if has_clip else 0.0,
                1 if has_clip else 0, float(mn), denom, int(layout), out[i:i + m].data_ptr(),
                _stream(device),
            ),
            "exaspim_gather_patches_as",
        )
    return out


# --- Helpers ---
def count_patches(img_shape, patch_shape, overlap):
    """
    Counts the number of patches within a 3D image for a given patch shape
    and overlap between the patches.

    Parameters
    ----------
    img_shape : Tuple[int]
        Shape (batch, channels, depth, height, width) of the image.
    patch_shape : Tuple[int]
        Shape of the 3D patch expected by the model.
    overlap : Tuple[int]
        Number of voxels in overlap between patches along each dimension.

    Returns
    -------
    int
        Number of patches.
    """
    assert len(img_shape) == 5, "Image must have shape (1, 1, D, H, W)"
    total = 1
    for axis_range in _start_ranges(img_shape[2:], patch_shape, overlap):
        total *= len(axis_range)
    return total


def generate_patch_starts(img_shape, patch_shape, overlap):
    """
    Generates starting coordinates for 3D patches extracted from an image
    tensor, based on specified patch size and overlap (z outermost, x
    fastest).

    Parameters
    ----------
    img_shape : Tuple[int]
        Shape (batch, channels, depth, height, width) of the image.
    patch_shape : Tuple[int]
        Shape of the 3D patch expected by the model.
    overlap : Tuple[int]
        Number of voxels in overlap between patches along each dimension.

    Returns
    -------
    Iterator[Tuple[int]]
        Starting coordinates of the image patches.
    """
    assert len(img_shape) == 5, "Image must have shape (1, 1, D, H, W)"
    yield from itertools.product(*_start_ranges(img_shape[2:], patch_shape, overlap))


def load_model(path, affinity_mode=True, device="cuda", *, compute_dtype="fp32"):
    """
    Loads a pretrained UNet model from a file.

    Parameters
    ----------
    path : str
        Path to the saved model weights (a state_dict written by the
        reference's trainer, train.py:286).
    affinity_mode : bool, optional
        If True, the model predicts affinities; if False, it predicts
        foreground-background. Default is True.
    device : str, optional
        Device to load the model onto. Default is "cuda" (the HIP device).
    compute_dtype : str, optional
        "fp32" (default), "bf16" or "fp16" arithmetic of the network kernels, or
        "auto": fp16 if this checkpoint, on the first batch of patches predict()
        gives it, stays inside half range and within 1e-3 of its own float32
        probabilities, float32 otherwise (UNet3D.resolve_compute_dtype).

    Returns
    -------
    model : torch.nn.Module
        UNet model loaded with weights and set to evaluation mode.
    """
    output_channels = 3 if affinity_mode else 1
    model = UNet3D(output_channels=output_channels, compute_dtype=compute_dtype)
    state = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(state)
    model.to(device)
    model.eval()
    return model


def to_tensor(arr, device="cuda"):
    """
    Converts a NumPy array to a float32 PyTorch tensor on a device, adding a
    channel axis until the array is 5-D.

    Parameters
    ----------
    arr : numpy.ndarray
        Array to be converted.
    device : str, optional
        Device to move the array to. Default is "cuda".

    Returns
    -------
    torch.Tensor
        Tensor on the device.
    """
    while arr.ndim < 5:
        arr = arr[:, np.newaxis, ...]
    return torch.tensor(arr).to(device, dtype=torch.float32)


# --- device-side building blocks (used by predict and by sharding.py) ---
class DeviceVolume:
    """
    A (block of a) 3D volume in device memory plus its place in the global
    volume.

    Attributes
    ----------
    tensor : torch.Tensor
        Device tensor of shape (D, H, W) (uint8, int16 holding uint16/int16
        bits, or float32).
    np_dtype : numpy.dtype
        Voxel dtype of the image as numpy sees it (what np.minimum and
        np.percentile of the reference operate on).
    storage_dtype : numpy.dtype
        Dtype the voxels have in device memory (uint8, uint16, int16, float32 or
        float64; wider integers and float64 travel as float32 when every value is
        known to be exactly representable, as float64 otherwise).
    block : _native.Block
        Local dims / global origin / global shape.
    """

    def __init__(self, tensor, np_dtype, origin=None, global_shape=None, storage_dtype=None):
        np_dtype = np.dtype(np_dtype)
        storage = np.dtype(storage_dtype) if storage_dtype is not None else _device_voxel_dtype(np_dtype)[0]
        if storage not in _VOX_CODES:
            raise TypeError(_unsupported(np_dtype))
        if tensor.dim() != 3 or not tensor.is_cuda:
            raise ValueError("DeviceVolume expects a 3-D tensor on a HIP device")
        if tensor.element_size() != storage.itemsize:
            raise ValueError("tensor element size does not match the voxel dtype")
        self.tensor = tensor.contiguous()
        self.np_dtype = np_dtype
        self.storage_dtype = storage
        self.vox_code = _VOX_CODES[storage]
        self.shape = tuple(int(v) for v in (global_shape or tensor.shape))
        self.block = _native.Block.make(tuple(tensor.shape), origin, self.shape)

    @classmethod
    def from_array(cls, img, device, clip=None):
        """Uploads a numpy array (3-D to 5-D, leading axes of length 1); "clip" is the
        brightness clip it will be used with (it can decide how wide integers and float64
        voxels travel, see _device_voxel_dtype)."""
        if isinstance(img, DeviceVolume):
            return img
        if isinstance(img, torch.Tensor):
            t = img
            while t.dim() > 3:
                if t.shape[0] != 1:
                    raise ValueError("leading image axes must have length 1")
                t = t[0]
            np_dtype = {
                torch.uint8: np.uint8, torch.int16: np.int16, torch.float32: np.float32,
                torch.float64: np.float64, getattr(torch, "uint16", None): np.uint16,
            }.get(t.dtype)
            if np_dtype is None:
                raise TypeError(f"tensor dtype {t.dtype} not supported")
            return cls(t.to(device), np_dtype)
        arr = np.asarray(img)
        while arr.ndim > 3:
            if arr.shape[0] != 1:
                raise ValueError("leading image axes must have length 1")
            arr = arr[0]
        if arr.ndim != 3:
            raise ValueError(f"expected a 3-D image, got shape {np.shape(img)}")
        storage, convert = _device_voxel_dtype(arr.dtype, whole=arr, clip=clip)
        if storage not in _VOX_CODES:
            raise TypeError(_unsupported(arr.dtype))
        return cls(_carrier(convert(arr)).to(device), arr.dtype, storage_dtype=storage)


def _unsupported(np_dtype):
    return (f"voxel dtype {np_dtype} not supported (8-, 16-, 32- and 64-bit integers, float32 and "
            "float64 are; 64-bit integers must be exactly representable in float64)")


class SlidingWindow:
    """
    Geometry of predict()'s sliding window over a volume of "shape".
    """

    def __init__(self, shape, patch_shape, overlap, trim):
        self.shape = tuple(int(v) for v in shape)
        self.patch_shape = tuple(int(v) for v in patch_shape)
        self.overlap = tuple(int(v) for v in overlap)
        self.trim = int(trim)
        if len(self.patch_shape) != 3 or len(self.overlap) != 3:
            raise ValueError("patch_shape and overlap must have three entries")
        if any(o >= p or o < 0 for o, p in zip(self.overlap, self.patch_shape)):
            raise ValueError("overlap must be in [0, patch_shape)")
        if self.trim < 0 or any(2 * self.trim >= p for p in self.patch_shape):
            raise ValueError("trim must satisfy 0 <= 2 * trim < patch_shape")
        self.window = _native.Window.make(self.patch_shape, self.overlap, self.trim)
        self.shape5 = (1, 1) + self.shape
        # The reference's stitch loop (inference.py:101-116) places a trimmed patch with
        # accum[s:e] += patch[:e - s], s = start + trim, e = min(s + out, dim). With
        # trim > overlap + 1 the last starts of an axis can have s > dim: e - s is negative,
        # patch[:e - s] is not empty while accum[s:e] is, and numpy raises. Same geometry,
        # same exception -- before any work is queued instead of half-way through.
        n_starts = [len(range(0, d - ov, ps - ov))
                    for d, ps, ov in zip(self.shape, self.patch_shape, self.overlap)]
        if self.trim > 0 and len(self.shape) == 3 and all(n_starts):   # no patch, no loop, no error
            for axis, (d, ps, ov) in enumerate(zip(self.shape, self.patch_shape, self.overlap)):
                out = ps - 2 * self.trim
                for s0 in reversed(range(0, d - ov, ps - ov)):
                    if s0 + self.trim <= d:
                        break
                    if s0 + self.trim < d + out:
                        raise ValueError(
                            "operands could not be broadcast together: the patch starting at "
                            f"{s0} on axis {axis} begins {s0 + self.trim - d} voxel(s) past the "
                            f"image (size {d}) after trimming {self.trim}; the reference's stitch "
                            "loop fails on this geometry (trim > overlap + 1)"
                        )

    def starts(self):
        """All patch starts in the reference's order."""
        return list(generate_patch_starts(self.shape5, self.patch_shape, self.overlap))


def _start_ranges(dims, patch_shape, overlap):
    """range(0, d - patch + stride, stride) per axis (inference.py:361-364)."""
    return [
        range(0, d - ps + (ps - ov), ps - ov)
        for d, ps, ov in zip(dims, patch_shape, overlap)
    ]


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


_WORKER_STREAMS = {}


def _worker_streams(device, n):
    """Side streams per device, created once (the model keeps one workspace per stream)."""
    pool = _WORKER_STREAMS.setdefault(str(device), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


def _effective_clip(np_dtype, brightness_clip, storage_dtype=None):
    """
    Applies numpy's promotion rules of np.minimum(img, brightness_clip)
    (inference.py:79); returns (clip, dtype of the clipped image): clip is None
    when no clip is given, otherwise the value the kernels compare against.
    An integer image with a fractional clip becomes float64, like in numpy.
    """
    np_dtype = np.dtype(np_dtype)
    if brightness_clip is None:
        return None, np_dtype
    probe = np.minimum(np.zeros(1, dtype=np_dtype), brightness_clip)  # may raise like numpy
    if probe.dtype == np_dtype:
        return np_dtype.type(brightness_clip), np_dtype
    if probe.dtype.kind in "iu" and np_dtype.kind in "iu":
        # a typed integer clip (np.int64(1000) on a uint16 image): numpy widens the image, every voxel
        # keeps its value, so the comparison can stay in the image's own dtype; the wider dtype only
        # decides the integer arithmetic inside np.percentile (value_dtype)
        info, c = np.iinfo(np_dtype), int(brightness_clip)
        if c < info.min:
            raise NotImplementedError(f"brightness_clip {brightness_clip!r} lies below every {np_dtype} voxel")
        return (None if c >= info.max else np_dtype.type(c)), probe.dtype
    if probe.dtype != np.float64:
        # a float16 / float32 typed clip on a narrow integer image: np.percentile and the
        # normalisation would then run in that float type's arithmetic
        raise NotImplementedError(
            f"brightness_clip {brightness_clip!r} promotes {np_dtype} voxels to {probe.dtype}"
        )
    clip = np.float64(brightness_clip)
    if np_dtype.kind in "iu" and clip < np.iinfo(np_dtype).min:
        raise NotImplementedError(f"brightness_clip {brightness_clip!r} lies below every {np_dtype} voxel")
    storage = np.dtype(storage_dtype) if storage_dtype is not None else _device_voxel_dtype(np_dtype)[0]
    if storage == np.float32 and np_dtype != np.float32 and np.float64(np.float32(clip)) != clip:
        # (predict() sends such a volume as float64 instead: _device_voxel_dtype(..., clip=))
        raise NotImplementedError(
            f"brightness_clip {brightness_clip!r} is not a float32 number (the volume travels as float32)"
        )
    return clip, np.dtype(np.float64)


def _histogram_into(hist, tensor, vox_code, clip, pass_index=0, prefix=0):
    """Adds the voxels of a device tensor to an int64 (65536,) device histogram."""
    if vox_code == _native.VOX_F64:
        _native.check(
            _native.lib().exaspim_histogram_wide(
                tensor.data_ptr(), vox_code, tensor.numel(),
                float(clip) if clip is not None else 0.0, 1 if clip is not None else 0,
                pass_index, int(prefix), hist.data_ptr(), _stream(tensor.device),
            ),
            "exaspim_histogram_wide",
        )
        return
    _native.check(
        _native.lib().exaspim_histogram(
            tensor.data_ptr(), vox_code, tensor.numel(),
            float(clip) if clip is not None else 0.0, 1 if clip is not None else 0,
            pass_index, prefix, hist.data_ptr(), _stream(tensor.device),
        ),
        "exaspim_histogram",
    )


def volume_histogram(volume, clip, pass_index=0, prefix=0):
    """Runs exaspim_histogram over a DeviceVolume; returns the int64 device tensor."""
    hist = torch.zeros(65536, dtype=torch.int64, device=volume.tensor.device)
    _histogram_into(hist, volume.tensor, volume.vox_code, clip, pass_index, prefix)
    return hist


def _key_to_f32(key):
    """Inverse of the order-preserving float32 key of the histogram kernel."""
    key = np.uint32(key)
    bits = (key & np.uint32(0x7FFFFFFF)) if key & np.uint32(0x80000000) else ~key
    return np.array([bits], dtype=np.uint32).view(np.float32)[0]


def _key_to_f64(key):
    """Inverse of the order-preserving float64 key of the wide histogram kernel."""
    key = int(key)
    bits = (key & 0x7FFFFFFFFFFFFFFF) if key >> 63 else (~key & 0xFFFFFFFFFFFFFFFF)
    return np.array([bits], dtype=np.uint64).view(np.float64)[0]


def _percentiles_from_histograms(histogram, storage_dtype, percentiles, value_dtype=None, clip=None):
    """
    np.percentile of the clipped volume from 65536-bin histograms:
    "histogram(pass_index, prefix)" returns the counts (numpy int64) over the
    whole volume; integer voxels need one pass, float32 voxels a second pass per
    16-bit key prefix an order statistic falls into. "value_dtype" is the dtype
    of the clipped image in numpy's eyes (it decides the arithmetic of
    np.percentile); with a fractional clip of an integer image every voxel above
    the clip sits in bin ceil(clip) and stands for the clip itself.
    """
    storage_dtype = np.dtype(storage_dtype)
    value_dtype = np.dtype(value_dtype) if value_dtype is not None else storage_dtype

    def as_value(v):
        if clip is not None and np.float64(v) > np.float64(clip):
            return value_dtype.type(clip)
        return value_dtype.type(v)

    if storage_dtype.kind in "ui":
        offset = 32768 if storage_dtype == np.int16 else 0
        stats = img_util.OrderStatistics(histogram(), lambda b: as_value(b - offset))
    elif storage_dtype == np.float64:
        # radix select on the order-preserving 64-bit key, 16 bits per level
        levels = {}

        def level(p, prefix):
            if (p, prefix) not in levels:
                levels[(p, prefix)] = img_util.OrderStatistics(histogram(p, prefix), lambda b: b)
            return levels[(p, prefix)]

        class _F64Stats:
            n = level(0, 0).n

            @staticmethod
            def kth(k):
                prefix, rank = 0, k
                for p in range(4):
                    b, rank = level(p, prefix).bin_of_rank(rank)
                    prefix = (prefix << 16) | b
                return as_value(_key_to_f64(prefix))

        stats = _F64Stats()
    else:
        coarse = img_util.OrderStatistics(histogram(0), lambda b: b)
        fine = {}

        class _F32Stats:
            n = coarse.n

            @staticmethod
            def kth(k):
                hi, within = coarse.bin_of_rank(k)
                if hi not in fine:
                    fine[hi] = img_util.OrderStatistics(histogram(1, hi), lambda b: b)
                lo, _ = fine[hi].bin_of_rank(within)
                return as_value(_key_to_f32((hi << 16) | lo))

        stats = _F32Stats()
    mn, mx = img_util.percentiles_from_statistics(stats, percentiles, value_dtype)
    return mn, mx


def volume_percentiles(volume, brightness_clip, percentiles, reduce_fn=None):
    """
    np.percentile(np.minimum(img, brightness_clip), percentiles) for a
    device-resident volume (inference.py:79, img_util.py:526), exactly.

    Parameters
    ----------
    volume : DeviceVolume
        Local block of the volume.
    brightness_clip : float
        Clip of inference.py:79 (None disables it).
    percentiles : Tuple[float]
        Lower and upper percentile.
    reduce_fn : Callable[[torch.Tensor], None], optional
        In-place sum of a histogram over all ranks (sharded volumes).

    Returns
    -------
    Tuple[numpy.float64]
        (mn, mx).
    """
    clip, value_dtype = _effective_clip(volume.np_dtype, brightness_clip, volume.storage_dtype)

    def histogram(pass_index=0, prefix=0):
        hist = volume_histogram(volume, clip, pass_index, prefix)
        if reduce_fn is not None:
            reduce_fn(hist)
        return hist.cpu().numpy()

    return _percentiles_from_histograms(histogram, volume.storage_dtype, percentiles, value_dtype, clip)


def _model_probabilities(model, inputs, trim=0, out=None):
    """sigmoid(model(inputs)) (inference.py:157-158) as a float32 device tensor;
    the "trim" voxels next to every patch face, which the caller discards
    (inference.py:161-162), are left undefined."""
    if isinstance(model, UNet3D):
        if FULL_PATCHES:  # measurement aid: compute the discarded margin too
            trim = 0
        return model.run(inputs, apply_sigmoid=True, trim=trim, out=out)
    with torch.no_grad():
        return torch.sigmoid(model(inputs)).to(torch.float32).contiguous()


def stitch_accumulate(pred, starts_dev, plan, accum, block):
    """accum += trimmed batch predictions (inference.py:99-116) on the device."""
    n = int(starts_dev.shape[0])
    # one launch takes n * trimmed depth <= 65535 grid rows; pieces in batch order add up
    # in the same order as one call
    piece = max(1, 65535 // max(1, plan.patch_shape[0] - 2 * plan.trim))
    for i in range(0, n, piece):
        m = min(piece, n - i)
        _native.check(
            _native.lib().exaspim_stitch_accumulate(
                pred[i:i + m].data_ptr(), starts_dev[i:i + m].data_ptr(), m,
                int(accum.shape[0]), plan.window, accum.data_ptr(), block,
                _stream(accum.device),
            ),
            "exaspim_stitch_accumulate",
        )


def stitch_finalize(accum, plan, block):
    """Divides by the per-voxel patch count (inference.py:120-125) on the device."""
    _native.check(
        _native.lib().exaspim_stitch_finalize(
            accum.data_ptr(), int(accum.shape[0]), plan.window, block, _stream(accum.device)
        ),
        "exaspim_stitch_finalize",
    )


def run_sliding_window(volume, model, plan, n_channels, batch_size, brightness_clip,
                       mn, mx, starts=None, accum=None, accum_block=None, verbose=False,
                       n_streams=1, pbar=None):
    """
    Runs every batch of the sliding window (inference.py:93-117): gather ->
    network -> sigmoid -> trimmed overlap-add into a device accumulator.
    Nothing here synchronises the device.

    Parameters
    ----------
    volume : DeviceVolume
        Input block in device memory (must contain every voxel the given
        starts read).
    model : torch.nn.Module
        Network on the same device.
    plan : SlidingWindow
        Window geometry over the GLOBAL volume.
    n_channels : int
        Output channels (3 affinities or 1 foreground map).
    batch_size : int
        Patches per batch.
    brightness_clip, mn, mx : float
        Pre-processing constants.
    n_streams : int, optional
        Batches in flight: gather + network of consecutive batches alternate
        between this many HIP streams (each with its own workspace); the stitch
        kernels stay on the caller's stream in batch order, so results do not
        depend on this number. Measured on MI355X (1024^3, 16-bit, batch 16): three
        streams are 5-7 % faster than one because the ramp-down of one kernel
        overlaps the next batch's work, which is why predict() asks for three;
        kernels of different batches then share the device and per-kernel
        timings lose their meaning, so this building block (and bench.py's
        roofline leg) defaults to 1. Moving only gather/stitch to a side stream
        gains nothing.
    starts : List[Tuple[int]], optional
        Patch starts to process (default: all of plan.starts()).
    accum : torch.Tensor, optional
        Existing float32 accumulator (n_channels, *accum_block.dims).
    accum_block : _native.Block, optional
        Placement of the accumulator (default: the volume's block).
    pbar : tqdm, optional
        Progress bar of the caller, advanced by the patches processed.

    Returns
    -------
    torch.Tensor
        The accumulator (sums, not yet divided).
    """
    device = volume.tensor.device
    clip, _ = _effective_clip(volume.np_dtype, brightness_clip, volume.storage_dtype)
    if starts is None:
        starts = plan.starts()
    if accum_block is None:
        accum_block = volume.block
    if accum is None:
        accum = torch.zeros((n_channels,) + tuple(accum_block.dims), dtype=torch.float32,
                            device=device)
    if len(starts) == 0:
        return accum
    if any(p % 16 for p in plan.patch_shape):
        # the reference's Up.forward fails in torch.cat for such sizes (unet3d.py:281-288)
        raise RuntimeError(
            "Sizes of tensors must match: patch_shape entries must be multiples of 16, "
            f"got {plan.patch_shape}"
        )
    starts_dev = torch.tensor(starts, dtype=torch.int32, device=device).reshape(-1, 3)
    # a kernel launch takes n * (patch depth + 2) <= 65535 grid rows: a caller's very large
    # batch of small patches runs as several engine batches (a patch's result does not depend
    # on the batch it travels in)
    batch_size = max(1, min(int(batch_size), 65535 // (max(plan.patch_shape) + 2)))
    own_pbar = pbar is None and verbose and tqdm is not None
    if own_pbar:
        pbar = tqdm(total=len(starts), desc="Predict")
    main = torch.cuda.current_stream(device)
    n_streams = max(1, min(int(n_streams), -(-len(starts) // batch_size)))
    workers = _worker_streams(device, n_streams) if n_streams > 1 else [main]
    for s in workers:
        s.wait_stream(main)  # volume, accumulator and starts are ready on the caller's stream
    if isinstance(model, UNet3D) and model.needs_resolution():
        # compute_dtype="auto": the first batch of real patches decides between fp16 and float32
        probe = _get_batch_inputs(volume, starts_dev[:batch_size], plan.patch_shape, device, clip=clip,
                                  mn=mn, mx=mx)
        model.resolve_compute_dtype(probe)
        del probe
    prepared_layout = None
    if isinstance(model, UNet3D) and not PLAIN_GATHER:
        prepared_layout = model.input_layout(device)    # (PLAIN_GATHER: tests hold the two paths to each other)
    for bi, i in enumerate(range(0, len(starts), batch_size)):
        batch = starts_dev[i:i + batch_size]
        worker = workers[bi % n_streams]
        with torch.cuda.stream(worker):
            if prepared_layout is not None:
                # the gather kernel writes the first convolution's operand layout directly
                inputs = _get_batch_inputs(volume, batch, plan.patch_shape, device, clip=clip,
                                           mn=mn, mx=mx, layout=prepared_layout)
                pred = model.run_prepared(
                    inputs, (int(batch.shape[0]),) + tuple(plan.patch_shape), apply_sigmoid=True,
                    trim=0 if FULL_PATCHES else plan.trim)
            else:
                inputs = _get_batch_inputs(volume, batch, plan.patch_shape, device, clip=clip,
                                           mn=mn, mx=mx)
                pred = _model_probabilities(model, inputs, plan.trim)
        if pred.shape[1] != n_channels:
            raise RuntimeError(
                f"model produced {pred.shape[1]} channels, expected {n_channels}"
            )
        if worker is not main:
            main.wait_stream(worker)
            pred.record_stream(main)
        stitch_accumulate(pred, batch, plan, accum, accum_block)
        if pbar is not None:
            pbar.update(int(batch.shape[0]))
    if own_pbar:
        pbar.close()
    return accum


# --- slab pipeline: out-of-core input, overlapped transfers ---
_COPY_STREAMS = {}
_PINNED_LOCK = threading.Lock()
_PINNED_FREE = {}               # (device, torch dtype) -> page-locked buffers no call is using
PINNED_SLOT_BYTES = 512 << 20   # a finished slab is cut so that one staging slot stays below this


def _checkout_pinned(device, count, numel, dtype):
    """
    Takes "count" host staging buffers of at least "numel" elements out of the pool of the
    device (page-locked, kept between calls so that a second predict() does not pay for
    hipHostMalloc again). A buffer belongs to ONE call from here until _return_pinned:
    concurrent predict() calls -- threads driving different GPUs, or the same one -- never
    share staging memory. If the runtime refuses to page-lock more memory the buffer is
    ordinary pageable memory (the download then blocks the copy stream, nothing else changes).
    """
    key = (str(device), dtype)
    with _PINNED_LOCK:
        free = _PINNED_FREE.setdefault(key, [])
        free.sort(key=lambda t: t.numel())
        taken = []
        while free and len(taken) < count and free[-1].numel() >= numel:
            taken.append(free.pop())
    while len(taken) < count:
        try:
            taken.append(torch.empty(numel, dtype=dtype, pin_memory=True))
        except RuntimeError:
            taken.append(torch.empty(numel, dtype=dtype))
    return taken


def _return_pinned(device, dtype, buffers):
    """Hands staging buffers back to the device's pool (pageable fall-backs are dropped)."""
    with _PINNED_LOCK:
        _PINNED_FREE.setdefault((str(device), dtype), []).extend(b for b in buffers if b.is_pinned())


def release_pinned_buffers():
    """Frees the page-locked staging buffers predict() keeps between calls."""
    with _PINNED_LOCK:
        _PINNED_FREE.clear()


def _copy_stream(device):
    """The stream the slab downloads run on (one per device, created once)."""
    key = str(device)
    with _PINNED_LOCK:
        if key not in _COPY_STREAMS:
            _COPY_STREAMS[key] = torch.cuda.Stream(device)
        return _COPY_STREAMS[key]


class _SlabDrain:
    """
    Moves finished output slabs off the device while later patch layers compute: a slab is
    written (and divided) into one of three device slots on the caller's stream, downloaded to
    a host staging slot on the copy stream, and handed to host threads from there.

    Parameters
    ----------
    device : torch.device
        The HIP device.
    slot_elems : int
        Elements of the largest slab handed to emit().
    half_out : bool
        Round slabs to IEEE half on the device before they leave it.
    threads : int
        Host threads that consume downloaded slabs.
    """

    N_SLOTS = 3

    def __init__(self, device, slot_elems, half_out, threads):
        self.device = device
        self.half_out = half_out
        self.main = torch.cuda.current_stream(device)
        self.copy_stream = _copy_stream(device)
        self.host_dtype = torch.float16 if half_out else torch.float32
        n = self.N_SLOTS
        self.dev_out = [torch.empty(slot_elems, dtype=torch.float32, device=device) for _ in range(n)]
        self.dev_half = ([torch.empty(slot_elems, dtype=torch.float16, device=device) for _ in range(n)]
                         if half_out else None)
        self.host = _checkout_pinned(device, n, slot_elems, self.host_dtype)
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(threads)))
        self.pending = [[] for _ in range(n)]
        self.n_emitted = 0

    def emit(self, shape, fill, consumers):
        """
        Queues one slab: "fill(out)" writes the final values into the zeroed float32 device
        tensor "out" of "shape"; "consumers(view)" returns the host jobs (callables) that read
        the downloaded numpy "view" of the same shape -- each runs on a pool thread once the
        download has finished, and the staging slot is reused only after all of them returned.
        """
        slot = self.n_emitted % self.N_SLOTS
        self.n_emitted += 1
        for f in self.pending[slot]:
            f.result()                   # the slot's previous slab has left the staging memory
        count = int(np.prod(shape))
        out = self.dev_out[slot][:count].view(shape)
        out.zero_()                      # planes no patch covers stay 0 (inference.py:120-125)
        fill(out)
        if self.half_out:
            out = export_half(out, self.dev_half[slot][:count]).view(shape)
        ready, done = torch.cuda.Event(), torch.cuda.Event()
        ready.record(self.main)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            self.host[slot][:count].view(shape).copy_(out, non_blocking=True)
            done.record(self.copy_stream)
        view = self.host[slot][:count].numpy().reshape(shape)

        def after_download(job):
            def run():
                done.synchronize()
                job()
            return run

        self.pending[slot] = [self.pool.submit(after_download(job)) for job in consumers(view)]

    def drain(self):
        """Waits until every queued slab has been consumed (re-raises a consumer's exception)."""
        for jobs in self.pending:
            for f in jobs:
                f.result()
        self.pending = [[] for _ in range(self.N_SLOTS)]

    def close(self):
        self.pool.shutdown(wait=True)
        _return_pinned(self.device, self.host_dtype, self.host)
        self.host = []


class _ArraySource:
    """read_block over anything that slices like a numpy array (ndarray, memmap,
    a zarr / N5 / TIFF-backed array as img_util.read returns, img_util.py:25-121)."""

    def __init__(self, arr):
        while len(arr.shape) > 3:
            if arr.shape[0] != 1:
                raise ValueError("leading image axes must have length 1")
            arr = arr[0]
        if len(arr.shape) != 3:
            raise ValueError(f"expected a 3-D image, got shape {tuple(arr.shape)}")
        self.arr = arr
        self.shape = tuple(int(v) for v in arr.shape)
        self.dtype = np.dtype(arr.dtype)

    def __call__(self, z0, z1):
        return np.asarray(self.arr[z0:z1])


def _device_voxel_dtype(np_dtype, whole=None, clip=None):
    """
    Voxel dtype the kernels read for an image of "np_dtype", and the function that converts a
    block to it. uint8 / uint16 / int16 / float32 / float64 travel as they are, int8 as int16.
    32- and 64-bit integers travel as float32 when the caller can show that every value is
    exactly representable ("whole": the entire image as an in-memory array) -- half the bytes --
    and as float64 otherwise (what the reference's arithmetic works in, img_util.py:526-531);
    float64 images likewise go as float32 only when "whole" proves them float32-exact. A
    brightness clip float32 cannot hold ("clip") sends them as float64 too. 64-bit integers that
    float64 cannot hold raise TypeError.
    """
    np_dtype = np.dtype(np_dtype)
    if np_dtype == np.int8:
        return np.dtype(np.int16), lambda block: block.astype(np.int16)
    if np_dtype.kind in "iuf" and np_dtype.itemsize in (4, 8) and np_dtype != np.float32:
        clip_ok = clip is None or np.float64(np.float32(clip)) == np.float64(clip)
        if whole is not None and clip_ok:
            as32 = np.asarray(whole).astype(np.float32)
            if np.array_equal(as32.astype(np_dtype), whole):
                def to_f32(block):
                    b32 = block.astype(np.float32)
                    if not np.array_equal(b32.astype(np_dtype), block):
                        raise TypeError(f"{np_dtype} volume is not exactly representable in float32")
                    return b32
                return np.dtype(np.float32), to_f32

        def to_f64(block):
            b64 = block.astype(np.float64)
            if np_dtype.kind in "iu" and np_dtype.itemsize == 8 and not np.array_equal(b64.astype(np_dtype), block):
                raise TypeError(f"{np_dtype} volume is not exactly representable in float64")
            return b64
        return np.dtype(np.float64), to_f64
    if np_dtype in _VOX_CODES:
        return np_dtype, lambda block: block
    return np_dtype, lambda block: block


def _carrier(block):
    """numpy block -> torch CPU tensor (uint16 rides as int16 bits)."""
    block = np.ascontiguousarray(block)
    if not block.flags.writeable:      # read-only memmap / zarr chunk: torch wants a writable buffer
        block = block.copy()
    if block.dtype == np.uint16:
        block = block.view(np.int16)
    return torch.from_numpy(block)


def predict_streaming(
    source,
    model,
    affinity_mode=True,
    batch_size=16,
    brightness_clip=1000,
    normalization_percentiles=(1, 99.9),
    patch_shape=(96, 96, 96),
    overlap=(32, 32, 32),
    trim=8,
    verbose=True,
    *,
    shape=None,
    dtype=None,
    write_block=None,
    keep_input_resident=None,
    n_streams=DEFAULT_STREAMS,
    copy_threads=4,
    timings=None,
    out_dtype=np.float32,
):
    """
    predict() for volumes that are read and written in z-slabs: the same
    results, bit for bit, without ever holding the whole input, the whole
    accumulator or the whole result on the device -- and, when "write_block"
    is given, not on the host either.

    The reference needs the whole image as one array (inference.py:79) and
    keeps a float64 normalised copy plus the float32 accumulators in host
    memory (22 B/voxel); its readers (img_util.py:25-121) hand out lazily
    chunked zarr / N5 / TIFF arrays that are sliced on demand. This function
    consumes exactly that interface:

    pass 1  every z-chunk is read once, uploaded and added to the device
            histogram (np.percentile of the clipped volume, exactly);
    pass 2  per patch layer k (all patches with z-start k * stride, the
            reference's z-major order, inference.py:368-397): upload the planes
            the layer adds to the rolling input slab, run gather -> U-Net ->
            sigmoid -> trimmed overlap-add into a one-layer accumulator that
            starts from the partial sums the previous layer left in the
            overlap band, then divide and emit the z-range no later layer
            touches. Finished slabs go device -> pinned memory on a copy stream
            and from there into the result (or to "write_block") on a few host
            threads while the next layer computes.

    Parameters
    ----------
    source : array-like or Callable[[int, int], numpy.ndarray]
        The image: anything that slices like a 3-D (or 1 x 1 x D x H x W) numpy
        array (ndarray, numpy.memmap, a zarr array), or a function
        read_block(z0, z1) returning planes [z0, z1) as a (z1 - z0, H, W) array,
        in which case "shape" and "dtype" are required.
    model : torch.nn.Module
        Model on a HIP device (see predict).
    affinity_mode, batch_size, brightness_clip, normalization_percentiles,
    patch_shape, overlap, trim, verbose
        As in predict (inference.py:29-40), same defaults.
    shape : Tuple[int], optional
        (D, H, W) of the image when "source" is a function.
    dtype : numpy.dtype, optional
        Voxel dtype when "source" is a function.
    write_block : Callable[[int, int, numpy.ndarray], None], optional
        Receives every finished output slab once, in z order, as
        write_block(z0, z1, block) with block float32 (C, z1 - z0, H, W) (or
        (z1 - z0, H, W) if not affinity_mode); the block is only valid during
        the call. Without it the full result array is returned.
    keep_input_resident : bool, optional
        Keep the uploaded image in HBM between the two passes instead of
        reading it twice. Default: True if it takes less than a quarter of
        the free device memory.
    n_streams : int, optional
        As in predict.
    copy_threads : int, optional
        Host threads that move finished slabs out of pinned memory. Default 4.
    timings : dict, optional
        Filled with the wall seconds of the phases ("upload_histogram",
        "layers", "drain"); measuring them synchronises the device twice.
    out_dtype : numpy.dtype, optional
        As in predict: numpy.float16 rounds every finished slab on the device,
        so the result array / the blocks handed to "write_block" are float16.

    Returns
    -------
    numpy.ndarray or None
        The prediction (see predict), or None when "write_block" is given.
    """
    device = next(model.parameters()).device
    if device.type != "cuda":
        raise RuntimeError(
            "predict (MI355X) has no CPU path: the model must be on a HIP device, "
            f"got {device}"
        )
    out_dtype = _checked_out_dtype(out_dtype)
    half_out = out_dtype == np.float16
    if callable(source) and not hasattr(source, "shape"):
        if shape is None or dtype is None:
            raise ValueError("a read_block function needs shape= and dtype=")
        read_block, vshape, src_dtype = source, tuple(int(v) for v in shape), np.dtype(dtype)
        if len(vshape) != 3:
            raise ValueError(f"expected a 3-D shape, got {vshape}")
    else:
        src = _ArraySource(source if hasattr(source, "shape") else np.asarray(source))
        read_block, vshape, src_dtype = src, src.shape, src.dtype
    whole = None
    if not callable(source) or hasattr(source, "shape"):
        arr = read_block.arr
        if isinstance(arr, np.ndarray) and not isinstance(arr, np.memmap):
            whole = arr      # an in-memory array can show that float32 carries it exactly
    vdtype, convert = _device_voxel_dtype(src_dtype, whole=whole, clip=brightness_clip)   # dtype in device memory
    if vdtype not in _VOX_CODES:
        raise TypeError(_unsupported(src_dtype))
    plan = SlidingWindow(vshape, patch_shape, overlap, trim)
    n_channels = 3 if affinity_mode else 1
    D, H, W = vshape
    pz = plan.patch_shape[0]
    stride = pz - plan.overlap[0]
    ranges = _start_ranges(vshape, plan.patch_shape, plan.overlap)
    yx_starts = list(itertools.product(*ranges[1:]))
    z_starts = list(ranges[0]) if yx_starts else []
    torch_dtype = _TORCH_VOXELS[vdtype]
    plane = H * W

    result = res4 = None
    if write_block is None:
        result = np.empty(((n_channels,) if affinity_mode else ()) + vshape, dtype=out_dtype)
        res4 = result.reshape((n_channels,) + vshape)

    def read(z0, z1):
        block = read_block(z0, z1)
        if tuple(block.shape) != (z1 - z0, H, W):
            raise ValueError(f"read_block({z0}, {z1}) returned shape {tuple(block.shape)}")
        return _carrier(convert(block))

    with torch.cuda.device(device):
        plane_bytes = plane * vdtype.itemsize
        if keep_input_resident is None:
            free, _ = torch.cuda.mem_get_info(device)
            keep_input_resident = D * plane_bytes <= free // 4
        chunk = max(1, min(D, max(stride, (256 << 20) // max(plane_bytes, 1))))

        # ---- pass 1: histogram of the clipped volume (inference.py:79, img_util.py:526) ----
        resident = None
        if keep_input_resident:
            resident = torch.empty((D, H, W), dtype=torch_dtype, device=device)
        clip, value_dtype = _effective_clip(src_dtype, brightness_clip, vdtype)
        loaded = [False]

        def chunks():
            for z0 in range(0, D, chunk):
                z1 = min(z0 + chunk, D)
                if resident is None:
                    yield read(z0, z1).to(device, non_blocking=True)
                else:
                    if not loaded[0]:
                        resident[z0:z1].copy_(read(z0, z1), non_blocking=True)
                    yield resident[z0:z1]
            loaded[0] = True

        def histogram(pass_index=0, prefix=0):
            hist = torch.zeros(65536, dtype=torch.int64, device=device)
            for part in chunks():
                _histogram_into(hist, part, _VOX_CODES[vdtype], clip, pass_index, prefix)
            return hist.cpu().numpy()

        t_phase = time.perf_counter()
        mn, mx = _percentiles_from_histograms(histogram, vdtype, normalization_percentiles, value_dtype, clip)
        if timings is not None:
            timings["upload_histogram"] = time.perf_counter() - t_phase
            t_phase = time.perf_counter()

        # ---- pass 2: patch layers ------------------------------------------------------
        band = pz - 2 * plan.trim - stride          # partial sums the next layer continues
        slab_d = min(pz, D)
        # deepest slab handed over at once: a layer finishes stride planes (the first one
        # stride + trim, the last one the patch depth - trim); anything deeper is cut, and so is
        # anything that would make a staging slot larger than PINNED_SLOT_BYTES
        max_out = min(D, max(stride + plan.trim, pz), max(1, PINNED_SLOT_BYTES // (n_channels * plane * 4)))
        # a sink is called from ONE thread, so the slabs arrive in z order, one at a time;
        # copies into the result array are split over several threads
        drain = _SlabDrain(device, n_channels * max_out * plane, half_out,
                           1 if write_block is not None else max(1, int(copy_threads)))
        acc_flat = [torch.empty(n_channels * slab_d * plane, dtype=torch.float32, device=device)
                    for _ in range(2)]
        in_slab = None
        if resident is None:
            in_slab = [torch.empty((slab_d, H, W), dtype=torch_dtype, device=device) for _ in range(2)]
        pbar = None
        if verbose and tqdm is not None:
            pbar = tqdm(total=len(z_starts) * len(yx_starts), desc="Predict")

        def acc_view(k):
            zs = z_starts[k]
            depth = min(zs + pz, D) - zs
            return acc_flat[k % 2][: n_channels * depth * plane].view(n_channels, depth, H, W)

        def consumers(z0, z1):
            """Host jobs for output planes [z0, z1) once they sit in staging memory."""
            def make(view):
                if write_block is not None:
                    return [lambda: write_block(z0, z1, view if affinity_mode else view[0])]
                # the copy into the pageable result is split over the threads
                pieces = max(1, min(int(copy_threads), z1 - z0))
                jobs = []
                for i in range(pieces):
                    a = z0 + (z1 - z0) * i // pieces
                    b = z0 + (z1 - z0) * (i + 1) // pieces
                    jobs.append(lambda a=a, b=b: np.copyto(res4[:, a:b], view[:, a - z0:b - z0]))
                return jobs
            return make

        def emit(z0, z1, fill):
            """Planes [z0, z1) are final: "fill(out)" writes the divided sums into the
            zeroed (C, z1 - z0, H, W) device slab, which then travels to the host."""
            drain.emit((n_channels, z1 - z0, H, W), fill, consumers(z0, z1))

        try:
            final_lo = 0
            for k, zs in enumerate(z_starts):
                hi = min(zs + pz, D)
                cur = acc_view(k)
                # -- input planes [zs, hi)
                if resident is not None:
                    volume = DeviceVolume(resident, src_dtype, storage_dtype=vdtype)
                else:
                    slab, old = in_slab[k % 2], in_slab[(k + 1) % 2]
                    have = 0
                    if k > 0:    # planes shared with the previous layer move on the device
                        zp = z_starts[k - 1]
                        have = max(0, min(zp + pz, D) - zs)
                        if have > 0:
                            slab[:have].copy_(old[zs - zp: zs - zp + have])
                    if hi - zs > have:
                        slab[have:hi - zs].copy_(read(zs + have, hi), non_blocking=True)
                    volume = DeviceVolume(slab[: hi - zs], src_dtype, (zs, 0, 0), vshape, storage_dtype=vdtype)
                # -- this layer's accumulator continues the previous layer's overlap band
                cur.zero_()
                if k > 0 and band > 0:
                    zp = z_starts[k - 1]
                    b0 = zs + plan.trim
                    b1 = min(b0 + band, D, zp + pz)
                    if b1 > b0:
                        cur[:, b0 - zs:b1 - zs].copy_(acc_view(k - 1)[:, b0 - zp:b1 - zp])
                blk = _native.Block.make((hi - zs, H, W), (zs, 0, 0), vshape)
                run_sliding_window(volume, model, plan, n_channels, batch_size, brightness_clip,
                                   mn, mx, starts=[(zs, y, x) for y, x in yx_starts], accum=cur,
                                   accum_block=blk, n_streams=n_streams, pbar=pbar)
                # -- planes no later layer touches
                final_hi = D if k + 1 == len(z_starts) else min(z_starts[k + 1] + plan.trim, D)
                for z0 in range(final_lo, final_hi, max_out):
                    z1 = min(z0 + max_out, final_hi)

                    def fill(out, z0=z0, z1=z1):
                        a, b = max(z0, zs), min(z1, hi)
                        if b > a:
                            out[:, a - z0:b - z0].copy_(cur[:, a - zs:b - zs])
                        stitch_finalize(out, plan, _native.Block.make((z1 - z0, H, W), (z0, 0, 0), vshape))

                    emit(z0, z1, fill)
                final_lo = max(final_lo, final_hi)
            # no patch fits (a dimension <= overlap): zeros, like the reference
            for z0 in range(final_lo, D, max_out):
                emit(z0, min(z0 + max_out, D), lambda out: None)
            if timings is not None:
                torch.cuda.synchronize(device)
                timings["layers"] = time.perf_counter() - t_phase
                t_phase = time.perf_counter()
            drain.drain()
            if timings is not None:
                timings["drain"] = time.perf_counter() - t_phase
        finally:
            drain.close()
            if pbar is not None:
                pbar.close()
    return result
