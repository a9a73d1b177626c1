"""
Seeded fuzz of predict()'s geometry on the GPU (-m gpu): random volume shapes,
patch shapes, overlaps, trims, batch sizes, voxel dtypes, clips and both output
modes, a quarter-width network -- against the CPU oracle (the restatement of
inference.py:79-126 pinned by the reference's golden vectors), and the three
ways through the package (device-resident, host array, streamed sink) against
each other bit for bit. The cases are fixed by the seed; a failure prints its
parameters.
"""

import os

import numpy as np
import pytest
import torch

from aind_exaspim_neuron_segmentation_amd.utils import synthetic

pytestmark = pytest.mark.gpu


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        patch = tuple(int(16 * rng.integers(1, 4)) for _ in range(3))               # 16, 32, 48
        overlap = tuple(int(rng.integers(0, p // 2 + 1)) for p in patch)
        trim = int(rng.integers(0, min(patch) // 4 + 1))
        # a volume from "no patch fits" over one ragged patch to a few patches per axis
        shape = tuple(int(rng.integers(max(2, o - 3), 2 * p + 25)) for p, o in zip(patch, overlap))
        out.append(dict(
            shape=shape, patch=patch, overlap=overlap, trim=trim, batch=int(rng.integers(1, 10)),
            affinity=bool(rng.integers(0, 3)), vox=["u16", "u8", "f32", "i16"][int(rng.integers(0, 4))],
            pct=[(1, 99.9), (0, 100), (5, 95)][int(rng.integers(0, 3))], seed=100 + i,
        ))
    return out


def _volume(case):
    v = synthetic.synth_volume(case["shape"], seed=case["seed"])
    if case["vox"] == "u8":
        return (v % 251).astype(np.uint8), 200
    if case["vox"] == "f32":
        return v.astype(np.float32) * 0.37 - 3.0, 520.5
    if case["vox"] == "i16":
        return (v.astype(np.int32) - 700).astype(np.int16), 900
    return v, 1000


# EXASPIM_FUZZ_CASES / EXASPIM_FUZZ_SEED widen the hunt by hand; the committed run is 24 cases
_N = int(os.environ.get("EXASPIM_FUZZ_CASES", "24"))
_SEED = int(os.environ.get("EXASPIM_FUZZ_SEED", "2026"))


@pytest.mark.parametrize("case", _cases(_N, seed=_SEED), ids=lambda c: "x".join(map(str, c["shape"])))
def test_predict_geometry_fuzz_vs_oracle(case):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from oracle import reference_path as oracle

    from aind_exaspim_neuron_segmentation_amd import inference
    from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D

    dev = torch.device("cuda:0")
    oc = 3 if case["affinity"] else 1
    sd = synthetic.synth_state_dict(oc, 0.25, seed=9)
    model = UNet3D(output_channels=oc, width_multiplier=0.25)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    model.to(dev).eval()
    vol, clip = _volume(case)
    kw = dict(affinity_mode=case["affinity"], batch_size=case["batch"], brightness_clip=clip,
              normalization_percentiles=case["pct"], patch_shape=case["patch"],
              overlap=case["overlap"], trim=case["trim"])
    print(case)
    try:
        want = oracle.predict(vol, oracle.OracleModel(sd), **kw)
    except ValueError as exc:
        # the reference's stitch loop cannot place a last patch that starts past the image
        # after trimming (trim > overlap + 1, inference.py:101-116): same exception here
        assert "broadcast" in str(exc)
        with pytest.raises(ValueError, match="broadcast"):
            inference.predict(vol, model, verbose=False, **kw)
        with pytest.raises(ValueError, match="broadcast"):
            inference.predict(vol, model, verbose=False, return_device_tensor=True, **kw)
        return
    got = inference.predict(vol, model, verbose=False, **kw)
    assert got.shape == want.shape and got.dtype == np.float32
    np.testing.assert_array_equal(got == 0, want == 0)
    err = np.abs(got - want).max() if got.size else 0.0
    assert err < 1e-5, err
    res = inference.predict(vol, model, verbose=False, return_device_tensor=True, **kw).cpu().numpy()
    np.testing.assert_array_equal(res, got)
    blocks = []
    inference.predict_streaming(lambda z0, z1: vol[z0:z1], model, verbose=False, shape=vol.shape,
                                dtype=vol.dtype, keep_input_resident=False,
                                write_block=lambda z0, z1, b: blocks.append(b.copy()), **kw)
    np.testing.assert_array_equal(np.concatenate(blocks, axis=1 if case["affinity"] else 0), got)
