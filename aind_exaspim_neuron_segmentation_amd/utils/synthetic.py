"""
Deterministic synthetic data for tests, golden fixtures and the benchmark.

Nothing here depends on torch's or numpy's RNG streams: every value is a pure
function of (seed, key, index) through the splitmix64 finaliser, so the build
container, the GPU box and the device-side generator
(``exaspim_synth_volume_u16`` in ``csrc/prepost.hip``) all produce identical
data.

* Volumes: ``uint16`` voxel = ``splitmix64(seed + global_linear_index) % 2000``
  (SURVEY.md section 8(d) "Synthetic input"); after the reference's brightness
  clip at 1000 this gives p1 = 19, p99.9 = 1000.
* Weights: one value stream per ``state_dict`` key (SURVEY.md section 8(d)
  "Synthetic weights"): conv W, b ~ U(+-1/sqrt(fan_in)); BatchNorm gamma ~
  U(0.5, 1.5), beta, running_mean ~ U(-0.2, 0.2), running_var ~ U(0.05, 0.55).
  Non-trivial running statistics matter: a freshly initialised BatchNorm is
  the identity and would hide BN-folding bugs.
"""

import zlib

import numpy as np

from aind_exaspim_neuron_segmentation_amd.machine_learning.spec import (
    unet_layer_specs,
)

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    """
    Applies the splitmix64 output function to an array of uint64 counters.

    Parameters
    ----------
    x : numpy.ndarray
        Array of dtype uint64.

    Returns
    -------
    numpy.ndarray
        Hashed values, dtype uint64.
    """
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_volume(shape, seed=0, origin=(0, 0, 0), global_shape=None):
    """
    Generates (a sub-block of) the synthetic uint16 volume.

    Parameters
    ----------
    shape : Tuple[int]
        Shape (D, H, W) of the block to generate.
    seed : int, optional
        Seed added to the global linear voxel index. Default is 0.
    origin : Tuple[int], optional
        Global coordinate of the block's first voxel. Default is (0, 0, 0).
    global_shape : Tuple[int], optional
        Shape of the whole volume the block is cut from. Default is "shape".

    Returns
    -------
    numpy.ndarray
        Block of dtype uint16 with values in [0, 2000).
    """
    gshape = tuple(global_shape) if global_shape is not None else tuple(shape)
    z = np.arange(origin[0], origin[0] + shape[0], dtype=np.uint64)
    y = np.arange(origin[1], origin[1] + shape[1], dtype=np.uint64)
    x = np.arange(origin[2], origin[2] + shape[2], dtype=np.uint64)
    lin = (
        z[:, None, None] * np.uint64(gshape[1]) + y[None, :, None]
    ) * np.uint64(gshape[2]) + x[None, None, :]
    h = splitmix64(lin + np.uint64(seed))
    return (h % np.uint64(2000)).astype(np.uint16)


def _uniform01(key, n, seed):
    """
    Returns n float64 values in [0, 1) from the stream named by (seed, key).
    """
    base = np.uint64(zlib.crc32(key.encode("utf-8"))) << np.uint64(32)
    with np.errstate(over="ignore"):
        offset = base + np.uint64(seed) * _GOLDEN
        h = splitmix64(np.arange(n, dtype=np.uint64) + offset)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def synth_state_dict(output_channels=3, width_multiplier=1, seed=1, trilinear=True):
    """
    Builds a synthetic UNet3D state_dict (numpy arrays) with the reference's
    keys in the reference's order (128 keys; 136 with "trilinear=False").

    Parameters
    ----------
    output_channels : int, optional
        Number of output channels of the head. Default is 3.
    width_multiplier : float, optional
        Channel width factor. Default is 1.
    seed : int, optional
        Seed of the value streams. Default is 1.
    trilinear : bool, optional
        False selects the ConvTranspose3d variant of the Up blocks. Default is
        True.

    Returns
    -------
    Dict[str, numpy.ndarray]
        float32 arrays (int64 scalars for "num_batches_tracked").
    """
    layers, (head_in, head_out) = unet_layer_specs(
        output_channels, trilinear, width_multiplier
    )
    sd = {}

    def uni(key, shape, lo, hi):
        n = int(np.prod(shape))
        u = _uniform01(key, n, seed)
        return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)

    for layer in layers:
        if layer[0] == "conv_transpose":
            _, prefix, cin, cout = layer
            bound = 1.0 / np.sqrt(float(cin))
            sd[f"{prefix}.weight"] = uni(f"{prefix}.weight", (cin, cout, 2, 2, 2), -bound, bound)
            sd[f"{prefix}.bias"] = uni(f"{prefix}.bias", (cout,), -bound, bound)
            continue
        _, prefix, cin, cmid, cout = layer
        for conv_idx, bn_idx, ci, co in ((0, 1, cin, cmid), (3, 4, cmid, cout)):
            bound = 1.0 / np.sqrt(27.0 * ci)
            k = f"{prefix}.{conv_idx}"
            sd[f"{k}.weight"] = uni(f"{k}.weight", (co, ci, 3, 3, 3), -bound, bound)
            sd[f"{k}.bias"] = uni(f"{k}.bias", (co,), -bound, bound)
            k = f"{prefix}.{bn_idx}"
            sd[f"{k}.weight"] = uni(f"{k}.weight", (co,), 0.5, 1.5)
            sd[f"{k}.bias"] = uni(f"{k}.bias", (co,), -0.2, 0.2)
            sd[f"{k}.running_mean"] = uni(f"{k}.running_mean", (co,), -0.2, 0.2)
            sd[f"{k}.running_var"] = uni(f"{k}.running_var", (co,), 0.05, 0.55)
            sd[f"{k}.num_batches_tracked"] = np.array(100, dtype=np.int64)
    bound = 1.0 / np.sqrt(float(head_in))
    sd["outc.conv.weight"] = uni(
        "outc.conv.weight", (head_out, head_in, 1, 1, 1), -bound, bound
    )
    sd["outc.conv.bias"] = uni("outc.conv.bias", (head_out,), -bound, bound)
    return sd
