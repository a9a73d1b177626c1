"""
GPU tests of the slab pipeline: predict() of a host
array (finished z-slabs downloaded while later patch layers compute) and
predict_streaming() over chunked sources (numpy.memmap, a read_block function,
a write_block sink) must reproduce the device-resident predict() bit for bit,
and their device footprint must not grow with the depth of the volume.
"""

import numpy as np
import pytest
import torch

from aind_exaspim_neuron_segmentation_amd.utils import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def make_model(dev, out_channels=3, seed=1, compute_dtype="fp32"):
    from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D

    sd = synthetic.synth_state_dict(out_channels, 1, seed=seed)
    model = UNet3D(output_channels=out_channels, compute_dtype=compute_dtype)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return model.to(dev).eval()


def resident(vol, model, **kw):
    """The whole-accumulator path: everything stays in HBM until one final copy."""
    from aind_exaspim_neuron_segmentation_amd import inference

    return inference.predict(vol, model, verbose=False, return_device_tensor=True, **kw).cpu().numpy()


@pytest.mark.parametrize("cdt", ["fp32", "fp16"])
def test_host_predict_equals_resident_predict_default_config_224(dev, cdt):
    """Reference defaults (96^3 patches, overlap 32, trim 8) on 224^3: three patch
    layers, 64-deep slabs, overlap bands carried from layer to layer; batches of 16
    are cut at layer boundaries (9 patches per layer), the resident path fills them
    across layers -- the bits must not depend on that."""
    from aind_exaspim_neuron_segmentation_amd import inference

    model = make_model(dev, compute_dtype=cdt)
    vol = synthetic.synth_volume((224, 224, 224), seed=2)
    want = resident(vol, model)
    got = inference.predict(vol, model, verbose=False)
    assert got.dtype == np.float32 and got.shape == (3, 224, 224, 224)
    np.testing.assert_array_equal(got, want)
    # 224 = 128 + 96: the last patch fits exactly, so both 8-voxel borders stay 0
    assert (got[:, :8] == 0).all() and (got[:, 216:] == 0).all() and (got[:, 200:216] != 0).any()


@pytest.mark.parametrize("shape,kw", [
    ((72, 88, 104), dict(patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4, batch_size=5)),
    ((90, 40, 56), dict(patch_shape=(32, 32, 32), overlap=(16, 16, 16), trim=2, batch_size=7)),
    ((64, 48, 40), dict(patch_shape=(32, 32, 32), overlap=(0, 0, 0), trim=8, batch_size=4)),   # gaps between outputs
    ((20, 40, 40), dict(patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4, batch_size=3)),   # one ragged layer
    ((8, 48, 40), dict(patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4, batch_size=3)),    # no patch fits
    ((48, 40, 8), dict(patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4, batch_size=3)),    # no patch fits (x)
])
def test_streaming_sources_and_sinks_equal_resident_predict(dev, tmp_path, shape, kw):
    from aind_exaspim_neuron_segmentation_amd import inference

    model = make_model(dev)
    vol = synthetic.synth_volume(shape, seed=7)
    want = resident(vol, model, **kw)

    # (a) numpy.memmap source, whole result returned
    path = str(tmp_path / "vol.u16")
    vol.tofile(path)
    mm = np.memmap(path, dtype=np.uint16, mode="r", shape=shape)
    got = inference.predict_streaming(mm, model, verbose=False, **kw)
    np.testing.assert_array_equal(got, want)

    # (b) read_block function, input not kept on the device, slabs handed to a sink
    reads, blocks = [], []

    def read_block(z0, z1):
        reads.append((z0, z1))
        return vol[z0:z1]

    def write_block(z0, z1, block):
        blocks.append((z0, z1, block.copy()))

    out = inference.predict_streaming(read_block, model, verbose=False, shape=shape, dtype=vol.dtype,
                                      write_block=write_block, keep_input_resident=False, **kw)
    assert out is None
    assert [b[0] for b in blocks] == sorted(b[0] for b in blocks)          # z order
    assert blocks[0][0] == 0 and blocks[-1][1] == shape[0]
    assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))           # each plane exactly once
    np.testing.assert_array_equal(np.concatenate([b[2] for b in blocks], axis=1), want)
    # the source is read twice (histogram pass + patch pass), never more
    planes_read = sum(z1 - z0 for z0, z1 in reads)
    assert planes_read <= 2 * shape[0]


def test_streaming_foreground_mode_and_float_volume(dev):
    """affinity_mode=False (one channel, 3-D slabs) on a float32 volume: the float
    percentile path needs a second histogram pass per 16-bit key prefix."""
    from aind_exaspim_neuron_segmentation_amd import inference

    model = make_model(dev, out_channels=1, seed=4)
    vol = (synthetic.synth_volume((56, 72, 40), seed=9).astype(np.float32) * 0.37)
    kw = dict(affinity_mode=False, patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4, batch_size=6,
              brightness_clip=250.5, normalization_percentiles=(3, 97))
    want = resident(vol, model, **kw)
    blocks = []
    inference.predict_streaming(lambda z0, z1: vol[z0:z1], model, verbose=False, shape=vol.shape,
                                dtype=vol.dtype, keep_input_resident=False,
                                write_block=lambda z0, z1, b: blocks.append(b.copy()), **kw)
    got = np.concatenate(blocks, axis=0)
    assert got.shape == want.shape == (56, 72, 40)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(inference.predict(vol[None, None], model, verbose=False, **kw), want)


def test_streaming_device_footprint_does_not_grow_with_depth(dev):
    """A run whose output would not fit one accumulator: with a read_block source, a
    write_block sink and keep_input_resident=False the device holds one input slab,
    two one-layer accumulators and three output slabs whatever the depth. Peak
    device memory of a 4x deeper volume stays the same (the full float32 result of
    the deep run alone, 3 x 1056 x 96 x 96 x 4 B = 117 MB, is several times the
    slab buffers), and the deep run equals the resident path."""
    from aind_exaspim_neuron_segmentation_amd import inference

    model = make_model(dev, compute_dtype="fp16")
    kw = dict(patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4, batch_size=9)

    def run(depth, check):
        vol = synthetic.synth_volume((depth, 96, 96), seed=3)
        csum = [0.0]
        parts = []

        def sink(z0, z1, b):
            csum[0] += float(b.sum(dtype=np.float64))
            if check:
                parts.append(b.copy())

        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_allocated(dev)
        inference.predict_streaming(lambda z0, z1: vol[z0:z1], model, verbose=False, shape=vol.shape,
                                    dtype=vol.dtype, keep_input_resident=False, write_block=sink, **kw)
        peak = torch.cuda.max_memory_allocated(dev) - base
        if check:
            np.testing.assert_array_equal(np.concatenate(parts, axis=1), resident(vol, model, **kw))
        return peak

    run(104, False)                       # warm-up: workspace and staging buffers exist afterwards
    shallow = run(264, False)
    deep = run(1056, True)
    print(f"peak device bytes above baseline: depth 264 -> {shallow}, depth 1056 -> {deep}")
    assert deep <= shallow + (1 << 20)
    assert deep < 3 * 1056 * 96 * 96 * 4 // 2


def test_half_precision_export(dev):
    """out_dtype=numpy.float16 (the reduced-precision export): the finished
    result rounded to IEEE half on the device -- bit for bit numpy's astype(float16) of the
    float32 result -- on the host path (slabs), through a write_block sink, and on the
    device-resident path; the consumer's astype(float32) (inference.py:223) is then within
    2.4e-4 of the float32 result."""
    from aind_exaspim_neuron_segmentation_amd import inference

    model = make_model(dev)
    vol = synthetic.synth_volume((72, 56, 43), seed=12)        # odd row length: unaligned tails
    kw = dict(patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4, batch_size=5)
    want32 = resident(vol, model, **kw)
    want16 = want32.astype(np.float16)
    got = inference.predict(vol, model, verbose=False, out_dtype=np.float16, **kw)
    assert got.dtype == np.float16 and got.shape == want32.shape
    np.testing.assert_array_equal(got.view(np.uint16), want16.view(np.uint16))
    assert np.abs(got.astype(np.float32) - want32).max() <= 2.4415e-4
    blocks = []
    inference.predict_streaming(vol, model, verbose=False, out_dtype="float16",
                                write_block=lambda z0, z1, b: blocks.append(b.copy()), **kw)
    assert all(b.dtype == np.float16 for b in blocks)
    np.testing.assert_array_equal(np.concatenate(blocks, axis=1).view(np.uint16), want16.view(np.uint16))
    t = inference.predict(vol, model, verbose=False, return_device_tensor=True, out_dtype=np.float16, **kw)
    assert t.dtype == torch.float16 and t.is_cuda
    np.testing.assert_array_equal(t.cpu().numpy().view(np.uint16), want16.view(np.uint16))
    # foreground mode: one channel, 3-D result
    m1 = make_model(dev, out_channels=1, seed=4)
    f32 = inference.predict(vol, m1, affinity_mode=False, verbose=False, **kw)
    f16 = inference.predict(vol, m1, affinity_mode=False, verbose=False, out_dtype=np.float16, **kw)
    assert f16.shape == vol.shape
    np.testing.assert_array_equal(f16.view(np.uint16), f32.astype(np.float16).view(np.uint16))
    with pytest.raises(TypeError):
        inference.predict(vol, model, verbose=False, out_dtype=np.uint8, **kw)
    # the export entry point on its own: every length class of the 8-wide kernel
    for n in (1, 7, 8, 9, 4099):
        src = torch.rand(n + 4, device=dev)[:n].contiguous()
        out = inference.export_half(src)
        assert torch.equal(out, src.to(torch.float16))


def test_concurrent_predict_calls_do_not_share_staging_memory(dev):
    """Two host threads call predict() at the same time (the pattern of one process driving
    several GPUs; here both on the one device, with different volumes and different models):
    every call checks its staging buffers out of the per-device pool, so neither overwrites
    the other's slabs. Each result equals the one the same call gives on its own."""
    import threading

    from aind_exaspim_neuron_segmentation_amd import inference

    kw = dict(patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4, batch_size=5)
    models = [make_model(dev, seed=1), make_model(dev, seed=2)]
    vols = [synthetic.synth_volume((120, 72, 88), seed=30), synthetic.synth_volume((136, 88, 72), seed=31)]
    alone = [inference.predict(v, m, verbose=False, **kw) for v, m in zip(vols, models)]
    got, errors = [None, None], []
    start = threading.Barrier(2)

    def work(i):
        try:
            torch.cuda.set_device(dev)
            start.wait(timeout=60)
            for _ in range(3):
                got[i] = inference.predict(vols[i], models[i], verbose=False, **kw)
        except Exception as exc:          # surfaced below
            errors.append(f"thread {i}: {type(exc).__name__}: {exc}")

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    for i in range(2):
        np.testing.assert_array_equal(got[i], alone[i])
    # the pool holds what the calls returned; releasing it empties it
    assert any(inference._PINNED_FREE.values())
    inference.release_pinned_buffers()
    assert not any(inference._PINNED_FREE.values())


def test_slabs_are_cut_to_the_staging_slot_size(dev, monkeypatch):
    """A staging slot stays below inference.PINNED_SLOT_BYTES whatever the plane size: slabs are
    cut thinner instead. With a tiny cap every emitted slab is one or two planes thick, the sink
    still sees every plane exactly once, in z order, and the result does not change by a bit."""
    from aind_exaspim_neuron_segmentation_amd import inference

    model = make_model(dev)
    kw = dict(patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4, batch_size=5)
    vol = synthetic.synth_volume((72, 56, 40), seed=8)
    want = resident(vol, model, **kw)
    monkeypatch.setattr(inference, "PINNED_SLOT_BYTES", 2 * 3 * 56 * 40 * 4)     # two planes
    inference.release_pinned_buffers()
    seen = []

    def sink(z0, z1, b):
        assert 0 < z1 - z0 <= 2 and b.shape == (3, z1 - z0, 56, 40)
        seen.append((z0, z1, b.copy()))

    inference.predict_streaming(vol, model, verbose=False, write_block=sink, **kw)
    assert [s[0] for s in seen] == sorted(s[0] for s in seen)
    assert seen[0][0] == 0 and seen[-1][1] == 72 and all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
    np.testing.assert_array_equal(np.concatenate([s[2] for s in seen], axis=1), want)
    np.testing.assert_array_equal(inference.predict(vol, model, verbose=False, **kw), want)
    inference.release_pinned_buffers()
