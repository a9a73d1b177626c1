"""
CPU-only checks of the C-ABI library and the host logic around it: the shared
object loads without a GPU, exports every symbol the header declares, packs
weights (BatchNorm folding + MFMA fragment order) correctly, and refuses to
compute without a device.
"""

import ctypes
import os
import re

import numpy as np
import pytest

from aind_exaspim_neuron_segmentation_amd import _native
from aind_exaspim_neuron_segmentation_amd.utils import img_util, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return _native.lib()


def test_header_symbols_are_exported(lib):
    header = open(os.path.join(ROOT, "include", "exaspim_affinity.h")).read()
    declared = set(re.findall(r"\b(exaspim_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.exaspim_abi_version() == 5


def test_param_count_matches_state_dict(lib):
    sd = synthetic.synth_state_dict(3, 1, seed=1)
    n = sum(v.size for k, v in sd.items() if not k.endswith("num_batches_tracked"))
    ch = _native.channels_array([32, 64, 128, 256, 512])
    assert lib.exaspim_unet_param_count(ch, 3, 0) == n == 12951267
    assert lib.exaspim_unet_param_count(_native.channels_array([4, 8, 16, 32, 64]), 1, 0) > 0
    # invalid widths are rejected with a message
    assert lib.exaspim_unet_param_count(_native.channels_array([32, 64, 128, 256, 500]), 3, 0) == 0
    assert "channels" in _native.last_error()


def _flat_params(sd):
    return np.concatenate(
        [v.reshape(-1).astype(np.float32) for k, v in sd.items()
         if not k.endswith("num_batches_tracked")]
    )


@pytest.mark.parametrize("dtype,code", [("f32", 0), ("bf16", 1), ("f16", 2)])
def test_pack_weights_folds_batchnorm_in_fragment_order(lib, dtype, code):
    widths = [4, 8, 16, 32, 64]  # exercises channel padding to 32
    sd = synthetic.synth_state_dict(3, 0.125, seed=3)
    params = _flat_params(sd)
    ch = _native.channels_array(widths)
    assert lib.exaspim_unet_param_count(ch, 3, 0) == params.size
    nbytes = lib.exaspim_unet_packed_bytes(ch, 3, code)
    packed = np.zeros(nbytes, np.uint8)
    rc = lib.exaspim_unet_pack_weights(ch, 3, code, params.ctypes.data, params.size,
                                       packed.ctypes.data, nbytes)
    _native.check(rc, "pack")

    # inc.0 block: float32 [27][32] folded weights then bias (256-byte aligned)
    w = sd["inc.double_conv.0.weight"].astype(np.float64).reshape(4, 27)
    b = sd["inc.double_conv.0.bias"].astype(np.float64)
    g = sd["inc.double_conv.1.weight"].astype(np.float64)
    beta = sd["inc.double_conv.1.bias"].astype(np.float64)
    mu = sd["inc.double_conv.1.running_mean"].astype(np.float64)
    var = sd["inc.double_conv.1.running_var"].astype(np.float64)
    s = g / np.sqrt(var + 1e-5)
    first_w = packed[: 27 * 32 * 4].view(np.float32).reshape(27, 32)
    np.testing.assert_array_equal(first_w[:, :4], (w * s[:, None]).T.astype(np.float32))
    assert not first_w[:, 4:].any()
    off_b = (27 * 32 * 4 + 255) // 256 * 256
    first_b = packed[off_b: off_b + 32 * 4].view(np.float32)
    np.testing.assert_array_equal(first_b[:4], ((b - mu) * s + beta).astype(np.float32))

    # inc.3 block (4 -> 4 channels, padded to 32 -> 32): check fragment order
    off_w = off_b + 256
    es = 4 if code == 0 else 2
    G = 16 // es
    KC = 2 * G
    nchunks = 32 // KC
    w3 = sd["inc.double_conv.3.weight"].astype(np.float64).reshape(4, 4, 27)
    s3 = sd["inc.double_conv.4.weight"].astype(np.float64) / np.sqrt(
        sd["inc.double_conv.4.running_var"].astype(np.float64) + 1e-5)
    raw = packed[off_w: off_w + 27 * 32 * 32 * es]
    if code == 0:
        vals = raw.view(np.float32).astype(np.float64)
    elif code == 1:
        vals = (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    else:
        vals = raw.view(np.float16).astype(np.float64)
    frag = vals.reshape(nchunks, 27, 1, 64, G)
    tol = {0: 0.0, 1: 2 ** -8, 2: 2 ** -10}[code]
    for c in range(nchunks):
        for lane in (0, 3, 31, 32, 35, 63):
            for j in range(G):
                ci = KC * c + G * (lane >> 5) + j
                co = lane & 31
                for t in (0, 13, 26):
                    want = w3[co, ci, t] * s3[co] if (ci < 4 and co < 4) else 0.0
                    got = frag[c, t, 0, lane, j]
                    if code == 0:
                        assert got == np.float32(want)
                    else:
                        assert abs(got - want) <= tol * abs(want) + 1e-30


def test_pack_conv_transpose_variant(lib):
    """EXASPIM_UP_CONVT: parameter order and [chunk][phase][tile][lane][4] fragments
    of the ConvTranspose3d(k=2, s=2) weights (unet3d.py:254-258)."""
    for wm, widths in ((1, [32, 64, 128, 256, 512]), (0.25, [8, 16, 32, 64, 128])):
        sd = synthetic.synth_state_dict(3, wm, seed=5, trilinear=False)
        assert len(sd) == 136
        params = _flat_params(sd)
        ch = _native.channels_array(widths)
        code = _native.UP_CONVT | _native.DT_F32
        assert lib.exaspim_unet_param_count(ch, 3, code) == params.size
        assert lib.exaspim_unet_param_count(ch, 3, 0) != params.size
        nbytes = lib.exaspim_unet_packed_bytes(ch, 3, code)
        packed = np.zeros(nbytes, np.uint8)
        _native.check(lib.exaspim_unet_pack_weights(ch, 3, code, params.ctypes.data, params.size,
                                                    packed.ctypes.data, nbytes), "pack")
        img = packed.view(np.float32)
        # up3.up: (Cin, Cout, 2, 2, 2) = (widths[2], widths[1], 2, 2, 2); locate its
        # fragment block through lane 0 of chunk 0 / phase 0 (ci = 0..3, co = 0)
        w = sd["up3.up.weight"]
        cin, cout = w.shape[:2]
        assert (cin, cout) == (widths[2], widths[1])
        cinp, coutp = -(-cin // 32) * 32, -(-cout // 32) * 32
        key = w[0:4, 0, 0, 0, 0]
        hits = [i for i in np.flatnonzero(img == key[0]) if np.array_equal(img[i:i + 4], key)]
        assert len(hits) == 1
        ntiles = coutp // 32
        frag = img[hits[0]: hits[0] + (cinp // 8) * 8 * ntiles * 64 * 4].reshape(cinp // 8, 8, ntiles, 64, 4)
        for c in (0, cinp // 8 - 1):
            for ph in (0, 5, 7):
                for n in range(ntiles):
                    for lane in (0, 31, 32, 63):
                        for j in range(4):
                            ci, co = 8 * c + 4 * (lane >> 5) + j, 32 * n + (lane & 31)
                            want = w[ci, co].reshape(8)[ph] if (ci < cin and co < cout) else 0.0
                            assert frag[c, ph, n, lane, j] == np.float32(want)
        # the bias follows, unfolded (a transposed conv has no BatchNorm)
        b = sd["up3.up.bias"]
        hits = [i for i in np.flatnonzero(img == b[0]) if np.array_equal(img[i:i + cout], b)]
        assert len(hits) == 1


def test_pack_rejects_wrong_sizes(lib):
    ch = _native.channels_array([32, 64, 128, 256, 512])
    buf = np.zeros(16, np.float32)
    rc = lib.exaspim_unet_pack_weights(ch, 3, 0, buf.ctypes.data, 16, buf.ctypes.data, 64)
    assert rc == -1 and "parameters" in _native.last_error()
    with pytest.raises(ValueError):
        _native.check(rc, "pack")


def test_create_without_device_fails_loudly(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ch = _native.channels_array([32, 64, 128, 256, 512])
    nbytes = lib.exaspim_unet_packed_bytes(ch, 3, 0)
    handle = ctypes.c_void_p()
    fake = ctypes.c_void_p(0x1000)
    rc = lib.exaspim_unet_create(ch, 3, 0, 0, fake, nbytes, ctypes.byref(handle))
    assert rc == -4 and "device" in _native.last_error()


def test_model_refuses_cpu_tensors():
    import torch

    from aind_exaspim_neuron_segmentation_amd import inference
    from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D

    model = UNet3D(output_channels=3).eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        model(torch.zeros(1, 1, 16, 16, 16))
    with pytest.raises(RuntimeError, match="no CPU path"):
        inference.predict(np.zeros((40, 40, 40), np.uint16), model, verbose=False)


def test_state_dict_keys_match_reference_layout():
    from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D

    model = UNet3D(output_channels=3)
    sd = synthetic.synth_state_dict(3, 1, seed=1)
    assert list(model.state_dict().keys()) == list(sd.keys())
    assert len(sd) == 128
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == sd[k].shape, k
    assert sum(p.numel() for p in model.parameters()) == 12946851


def test_patch_helpers_match_golden(golden):
    from aind_exaspim_neuron_segmentation_amd import inference

    g = golden("g1_patch_starts.npz")
    for i in range(int(g["n_cases"])):
        vol = tuple(int(v) for v in g[f"case{i}_vol"])
        ps = tuple(int(v) for v in g[f"case{i}_patch"])
        ov = tuple(int(v) for v in g[f"case{i}_overlap"])
        shape5 = (1, 1) + vol
        assert inference.count_patches(shape5, ps, ov) == int(g[f"case{i}_count"])
        starts = np.array(list(inference.generate_patch_starts(shape5, ps, ov)),
                          dtype=np.int64).reshape(-1, 3)
        if vol == (512, 512, 512):
            starts = starts[[0, 1, 7, 8, 63, 64, 510, 511]]
        np.testing.assert_array_equal(starts, g[f"case{i}_starts"])
    with pytest.raises(AssertionError):
        inference.count_patches((96, 96, 96), (96,) * 3, (32,) * 3)
    sl = img_util.get_patch_slices((64, 0, 128), (96, 96, 96), (130, 50, 224))
    g2 = golden("g2b_padding.npz")
    np.testing.assert_array_equal(np.array([[s.start, s.stop] for s in sl]), g2["slices"])


def test_percentiles_from_histogram_match_numpy(golden):
    g = golden("g2_normalize.npz")
    vol = synthetic.synth_volume((40, 48, 56), seed=3)
    cases = {
        "u16_clip1000": np.minimum(vol, 1000),
        "u16_noclip": vol,
        "u8": (vol % 251).astype(np.uint8),
        "i16": (vol.astype(np.int32) - 1000).astype(np.int16),
        "const": np.full((8, 8, 8), 7, dtype=np.uint16),
    }
    for name, arr in cases.items():
        offset = 32768 if arr.dtype == np.int16 else 0
        hist = np.bincount(arr.astype(np.int64).ravel() + offset, minlength=65536)
        stats = img_util.OrderStatistics(hist, lambda b, d=arr.dtype, o=offset: d.type(b - o))
        for pct_name, pct in (("default", (1, 99.9)), ("alt", (0.5, 75.25))):
            got = img_util.percentiles_from_statistics(stats, pct, arr.dtype)
            np.testing.assert_array_equal(got, g[f"{name}_{pct_name}_mnmx"])


def test_reflect_index_matches_numpy_pad():
    for n in (1, 2, 3, 17, 33):
        for p in (0, 1, n - 1, n, 2 * n + 3, 63):
            if p < 0:
                continue
            a = np.arange(n)
            want = np.pad(a, (0, p), mode="reflect")
            got = [img_util.reflect_index(j, n) for j in range(n + p)]
            np.testing.assert_array_equal(got, want)


def test_validation_tiling_matches_reference_golden(golden):
    """generate_patch_starts with zero overlap + patch centres + is_contained(buffer=64):
    the tiling ValidateDataset builds its examples from (data_handling.py:402-413), against
    centres produced by the reference's own helpers."""
    from aind_exaspim_neuron_segmentation_amd import inference
    from aind_exaspim_neuron_segmentation_amd.utils import img_util

    g = golden("g8_validation_tiling.npz")
    for i in range(int(g["n_cases"])):
        vol = tuple(int(v) for v in g[f"case{i}_vol"])
        ps = tuple(int(v) for v in g[f"case{i}_patch"])
        starts = list(inference.generate_patch_starts((1, 1) + vol, ps, (0, 0, 0)))
        np.testing.assert_array_equal(np.array(starts, dtype=np.int64).reshape(-1, 3), g[f"case{i}_starts"])
        assert inference.count_patches((1, 1) + vol, ps, (0, 0, 0)) == len(starts)
        centers = [[v + s // 2 for v, s in zip(st, ps)] for st in starts]
        kept = [c for c in centers if img_util.is_contained(c, vol, buffer=64)]
        np.testing.assert_array_equal(np.array(kept, dtype=np.int64).reshape(-1, 3),
                                      g[f"case{i}_centers_kept"])


def test_is_contained_is_two_sided_like_the_reference():
    """img_util.py:472-474 requires 0 <= v + buffer < s AND 0 <= v - buffer < s per axis; the two
    one-sided halves agree with it only for buffer >= 0. Known answers worked out from those two lines."""
    from aind_exaspim_neuron_segmentation_amd.utils import img_util

    def rule(voxel, shape, buffer):
        return (all(0 <= v + buffer < s for v, s in zip(voxel, shape))
                and all(0 <= v - buffer < s for v, s in zip(voxel, shape)))

    assert img_util.is_contained((0, 0, 0), (10, 10, 10), buffer=-1) is False      # v + buffer = -1
    assert img_util.is_contained((9, 9, 9), (10, 10, 10), buffer=-1) is False      # v - buffer = 10
    assert img_util.is_contained((5, 5, 5), (10, 10, 10), buffer=-1) is True
    assert img_util.is_contained((5, 5, 5), (10, 10, 10), buffer=5) is False       # v + buffer = 10
    assert img_util.is_contained((5, 5, 5), (11, 11, 11), buffer=5) is True
    rng = np.random.default_rng(3)
    for _ in range(2000):
        shape = tuple(int(v) for v in rng.integers(1, 12, 3))
        voxel = tuple(int(v) for v in rng.integers(-3, 14, 3))
        buffer = int(rng.integers(-4, 5))
        assert img_util.is_contained(voxel, shape, buffer) == rule(voxel, shape, buffer)


def test_dtype_rules_and_fractional_clip_percentiles_on_the_host():
    """Host logic behind the voxel dtypes predict() takes: storage dtype per image dtype,
    numpy's np.minimum promotion, and np.percentile rebuilt from a 65536-bin histogram in
    which the voxels above a fractional clip sit in bin ceil(clip) (csrc/prepost.hip)."""
    import numpy as np
    from aind_exaspim_neuron_segmentation_amd import inference as inf

    for dt, want in ((np.uint8, np.uint8), (np.int8, np.int16), (np.uint16, np.uint16), (np.int16, np.int16),
                     (np.float32, np.float32), (np.int32, np.float32), (np.uint32, np.float32),
                     (np.int64, np.float32), (np.uint64, np.float32), (np.float64, np.float32)):
        block = (np.arange(24).reshape(2, 3, 4) % 7).astype(dt)
        # the whole image is at hand and float32 holds it: the narrow carrier
        storage, convert = inf._device_voxel_dtype(dt, whole=block)
        assert storage == np.dtype(want)
        np.testing.assert_array_equal(convert(block).astype(np.float64), block.astype(np.float64))
        # nothing known about the values (a chunked source): wide dtypes travel as float64
        storage, convert = inf._device_voxel_dtype(dt)
        assert storage == (np.dtype(np.float64) if np.dtype(dt).itemsize >= 4 and dt != np.float32 else np.dtype(want))
        np.testing.assert_array_equal(convert(block).astype(np.float64), block.astype(np.float64))
    # values float32 cannot hold: float64 carrier (the reference works in float64, img_util.py:526-531)
    wide = np.array([[[(1 << 40) + 1, 3]]], dtype=np.int64)
    storage, convert = inf._device_voxel_dtype(np.int64, whole=wide)
    assert storage == np.dtype(np.float64) and convert(wide)[0, 0, 0] == float((1 << 40) + 1)
    frac = np.array([[[0.1, 0.25]]])
    assert inf._device_voxel_dtype(np.float64, whole=frac)[0] == np.dtype(np.float64)
    assert inf._device_voxel_dtype(np.float64, whole=frac * 0 + 0.25)[0] == np.dtype(np.float32)
    # ... and so does a clip float32 cannot hold, whatever the values
    assert inf._device_voxel_dtype(np.int32, whole=np.zeros((1, 1, 2), np.int32), clip=0.1)[0] == np.dtype(np.float64)
    with pytest.raises(TypeError):      # beyond 2^53 float64 is not exact either
        inf._device_voxel_dtype(np.int64)[1](np.array([[[(1 << 60) + 1]]], dtype=np.int64))
    assert inf._device_voxel_dtype(np.complex64)[0] not in inf._VOX_CODES
    # the order-preserving 64-bit key of the wide histogram and its inverse
    for v in (0.0, -0.0, 1.0, -1.0, 0.1, 1e300, -1e-300, 1000.5):
        u = int(np.array([v]).view(np.uint64)[0])
        key = (~u & 0xFFFFFFFFFFFFFFFF) if u >> 63 else (u | (1 << 63))
        assert inf._key_to_f64(key) == v and np.signbit(inf._key_to_f64(key)) == np.signbit(v)

    assert inf._effective_clip(np.uint16, 1000) == (np.uint16(1000), np.dtype(np.uint16))
    assert inf._effective_clip(np.uint16, None) == (None, np.dtype(np.uint16))
    assert inf._effective_clip(np.uint16, 1000.5) == (np.float64(1000.5), np.dtype(np.float64))
    assert inf._effective_clip(np.float32, 123.5) == (np.float32(123.5), np.dtype(np.float32))
    with pytest.raises(OverflowError):
        inf._effective_clip(np.uint8, 1000)
    with pytest.raises(NotImplementedError):      # an int32 image forced onto the float32 carrier; 0.1 is not a float32 number
        inf._effective_clip(np.int32, 0.1, np.float32)
    assert inf._effective_clip(np.int32, 0.1, np.float64) == (np.float64(0.1), np.dtype(np.float64))
    # typed integer clips widen the image in np.minimum; the voxels and the comparison stay as they are
    assert inf._effective_clip(np.uint16, np.int64(1000)) == (np.uint16(1000), np.dtype(np.int64))
    assert inf._effective_clip(np.int16, np.int32(40000)) == (None, np.dtype(np.int32))
    assert inf._effective_clip(np.uint8, np.int16(255)) == (None, np.dtype(np.int16))
    with pytest.raises(NotImplementedError):      # everything would become the clip
        inf._effective_clip(np.uint16, np.int32(-5))
    with pytest.raises(NotImplementedError):      # (a fractional one too: no bin stands for it)
        inf._effective_clip(np.uint16, -3.5)
    assert inf._effective_clip(np.int16, -3.5) == (np.float64(-3.5), np.dtype(np.float64))
    with pytest.raises(NotImplementedError):      # float32 arithmetic in np.percentile and normalize
        inf._effective_clip(np.uint16, np.float32(5))
    # an int16 image whose order statistics lie more than 32767 apart: numpy's own int16 lerp wraps,
    # the widened one does not -- value_dtype carries that difference
    wide = np.array([-30000, -30000, 30000, 30000], dtype=np.int16)
    for c in (np.int32(40000), 32767):
        cc, vdt = inf._effective_clip(np.int16, c)
        st = img_util.OrderStatistics(np.bincount(wide.astype(np.int64) + 32768, minlength=65536),
                                      lambda b: vdt.type(b - 32768))
        np.testing.assert_array_equal(img_util.percentiles_from_statistics(st, (40, 60), vdt),
                                      np.percentile(np.minimum(wide, c), (40, 60)))

    vol = synthetic.synth_volume((24, 20, 28), seed=5)
    for clip in (1000.5, 37.25, 5000.75):
        bins = np.where(vol.astype(np.float64) > clip, int(np.ceil(clip)), vol).astype(np.int64)
        hist = np.bincount(bins.ravel(), minlength=65536)
        c, vdt = inf._effective_clip(vol.dtype, clip)
        got = inf._percentiles_from_histograms(lambda *a: hist, vol.dtype, (1, 99.9), vdt, c)
        np.testing.assert_array_equal(np.array(got), np.percentile(np.minimum(vol, clip), (1, 99.9)))


def test_float64_percentiles_from_four_16_bit_passes_on_the_host():
    """np.percentile of a float64 (or wide-integer) volume from the wide histogram's passes
    (exaspim_histogram_wide: 16 bits of an order-preserving 64-bit key per pass), emulated here
    with numpy: equal to numpy bit for bit, clip or no clip, negative values, ties."""
    from aind_exaspim_neuron_segmentation_amd import inference as inf

    rng = np.random.default_rng(11)

    def passes(vol64, clip):
        v = np.minimum(vol64, clip) if clip is not None else vol64
        u = v.view(np.uint64)
        key = np.where(u >> np.uint64(63) != 0, ~u, u | np.uint64(1 << 63))
        calls = []

        def histogram(p=0, prefix=0):
            calls.append((p, prefix))
            k = key if p == 0 else key[(key >> np.uint64(64 - 16 * p)) == np.uint64(prefix)]
            return np.bincount(((k >> np.uint64(48 - 16 * p)) & np.uint64(0xFFFF)).astype(np.int64), minlength=65536)
        return histogram, calls

    cases = [
        (rng.normal(500.0, 300.0, 4000), 1000.0, np.float64, (1, 99.9)),
        (rng.normal(0.0, 1e-3, 3000), None, np.float64, (0, 100)),
        (np.round(rng.normal(0, 3, 5000)), 2.5, np.float64, (5, 50)),                 # ties, fractional clip
        (rng.integers(-(1 << 40), 1 << 40, 2000).astype(np.float64), None, np.int64, (1, 99.9)),   # wide integers
        (rng.integers(0, 1 << 31, 2000).astype(np.float64), 1 << 30, np.uint32, (2.5, 97.5)),
    ]
    for vol64, clip, value_dtype, pcts in cases:
        histogram, calls = passes(np.ascontiguousarray(vol64), clip)
        got = inf._percentiles_from_histograms(histogram, np.float64, pcts, value_dtype, clip)
        img = vol64.astype(value_dtype)
        want = np.percentile(np.minimum(img, clip) if clip is not None else img, pcts)
        np.testing.assert_array_equal(np.array(got), want)
        assert calls[0] == (0, 0) and all(p <= 3 for p, _ in calls) and len(calls) <= 1 + 3 * 4


def test_sliding_window_raises_where_the_reference_stitch_loop_raises():
    """inference.py:101-116 places a trimmed patch with accum[s:e] += patch[:e - s],
    s = start + trim, e = min(s + out, dim). A last start with s > dim makes e - s negative:
    patch[:e - s] is non-empty while accum[s:e] is empty and numpy raises ValueError
    (possible only if trim > overlap + 1). SlidingWindow raises for exactly those geometries:
    checked against the slice arithmetic itself on 5000 random axes."""
    from aind_exaspim_neuron_segmentation_amd import inference

    rng = np.random.default_rng(0)
    n_raise = 0
    for _ in range(5000):
        p = int(rng.integers(4, 40))
        ov = int(rng.integers(0, p))
        trim = int(rng.integers(0, (p - 1) // 2 + 1))
        d = int(rng.integers(1, 120))
        if 2 * trim >= p:
            continue
        out = p - 2 * trim
        ref_raises = False
        for s0 in range(0, d - p + (p - ov), p - ov):
            s = max(s0 + trim, 0)
            e = min(s + out, d)
            ref_raises |= len(range(*slice(s, e).indices(d))) != len(range(*slice(0, e - s).indices(out)))
        try:
            inference.SlidingWindow((d, d, d), (p, p, p), (ov, ov, ov), trim)
            ours = False
        except ValueError as exc:
            ours = "broadcast" in str(exc)
        assert ours == ref_raises, (p, ov, trim, d)
        n_raise += ref_raises
    assert n_raise > 50
    # the reference's defaults are far from it
    inference.SlidingWindow((1024, 1024, 1024), (96, 96, 96), (32, 32, 32), 8)
    # an axis without any patch start: the loop never runs, nothing raises (the result is zeros)
    with pytest.raises(ValueError, match="broadcast"):
        inference.SlidingWindow((86, 40, 3), (48, 32, 16), (9, 13, 2), 4)
    inference.SlidingWindow((86, 12, 3), (48, 32, 16), (9, 13, 2), 4)
