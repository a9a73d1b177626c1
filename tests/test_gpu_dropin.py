"""
GPU tests of the drop-in surface north_star names: a checkpoint written the way
the reference's trainer writes it (train.py:286: torch.save(model.state_dict()))
-> inference.load_model(path) -> inference.predict(img, model), against the
golden outputs of the reference itself (tests/golden/g5, g6); to_tensor; the
generic nn.Module branch of predict; the C-ABI stub printed in INTEGRATION.md
section 2 executed verbatim inside a reference-shaped _predict_batch loop; and
the numeric range of the fp16 storage mode.
"""

import os
import re

import numpy as np
import pytest
import torch

from aind_exaspim_neuron_segmentation_amd.utils import synthetic

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def oracle():
    from oracle import reference_path

    return reference_path


def save_checkpoint(path, out_channels, seed, trilinear=True):
    """A state_dict file as train.py:286 writes it (plain torch.save of tensors,
    including the BatchNorm num_batches_tracked entries)."""
    sd = synthetic.synth_state_dict(out_channels, 1, seed=seed, trilinear=trilinear)
    tensors = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    for k in list(tensors):
        if k.endswith("running_var"):
            tensors[k.replace("running_var", "num_batches_tracked")] = torch.tensor(7, dtype=torch.long)
    torch.save(tensors, path)
    return sd


# ---------------------------------------------------- load_model -> predict ---
def test_load_model_then_predict_default_config_vs_reference_golden(dev, golden, tmp_path):
    """inference.py:400-424 + 29-126 with every default: device="cuda", fp32."""
    from aind_exaspim_neuron_segmentation_amd import inference

    path = str(tmp_path / "UNet3d-20260101-1-0.9000.pth")
    save_checkpoint(path, 3, seed=1)
    model = inference.load_model(path)
    assert isinstance(model, torch.nn.Module) and not model.training
    assert next(model.parameters()).device.type == "cuda"
    assert len(model.state_dict()) == 128          # the reference's 128 keys, strict load
    g = golden("g6_default_160.npz")
    vol = synthetic.synth_volume((160, 160, 160), seed=0)
    got = inference.predict(vol, model, batch_size=8, verbose=False)
    err = np.abs(got[:, ::5, ::5, ::5] - g["pred_sub"]).max()
    print(f"load_model -> predict 160^3 vs reference: {err:.3e}")
    assert got.dtype == np.float32 and got.shape == (3, 160, 160, 160)
    assert err < 5e-6
    assert np.abs(got[:, 80, 81, :] - g["pred_line"]).max() < 5e-6


def test_load_model_foreground_mode_vs_reference_golden(dev, golden, tmp_path):
    """affinity_mode=False: one output channel, 3-D result (inference.py:419,126)."""
    from aind_exaspim_neuron_segmentation_amd import inference

    path = str(tmp_path / "fg.pth")
    save_checkpoint(path, 1, seed=4)
    model = inference.load_model(path, affinity_mode=False, device="cuda:0")
    g = golden("g5_fullwidth_small.npz")
    vol = synthetic.synth_volume((72, 40, 56), seed=11)
    got = inference.predict(vol, model, affinity_mode=False, batch_size=5, patch_shape=(32, 32, 32),
                            overlap=(16, 16, 16), trim=2, verbose=False)
    assert got.shape == (72, 40, 56)
    err = np.abs(got[::2, ::2, ::2] - g["pred_fg"]).max()
    print(f"load_model(affinity_mode=False) -> predict vs reference: {err:.3e}")
    assert err < 5e-6


@pytest.mark.parametrize("cdt,tol", [("fp16", 1e-3), ("bf16", 4e-3)])
def test_load_model_16bit_compute_dtype(dev, golden, tmp_path, cdt, tol):
    from aind_exaspim_neuron_segmentation_amd import inference

    path = str(tmp_path / "m.pth")
    save_checkpoint(path, 3, seed=1)
    model = inference.load_model(path, compute_dtype=cdt)
    g = golden("g6_default_160.npz")
    vol = synthetic.synth_volume((160, 160, 160), seed=0)
    got = inference.predict(vol, model, batch_size=8, verbose=False)
    err = np.abs(got[:, ::5, ::5, ::5] - g["pred_sub"])
    print(f"load_model({cdt}) -> predict vs reference: max {err.max():.3e} mean {err.mean():.3e}")
    assert err.max() < tol


def test_load_model_rejects_wrong_checkpoint(dev, tmp_path):
    from aind_exaspim_neuron_segmentation_amd import inference

    path = str(tmp_path / "fg.pth")
    save_checkpoint(path, 1, seed=4)
    with pytest.raises(RuntimeError, match="size mismatch|Missing|Unexpected"):
        inference.load_model(path, affinity_mode=True)   # 3-channel head vs 1-channel file


# ------------------------------------------------------------------ to_tensor ---
def test_to_tensor(dev):
    """inference.py:427-446: channel axes until 5-D, float32, on the device."""
    from aind_exaspim_neuron_segmentation_amd import inference

    arr = np.arange(2 * 4 * 5 * 6, dtype=np.uint16).reshape(2, 4, 5, 6)
    t = inference.to_tensor(arr)
    assert t.shape == (2, 1, 4, 5, 6) and t.dtype == torch.float32 and t.is_cuda
    np.testing.assert_array_equal(t.cpu().numpy()[:, 0], arr.astype(np.float32))
    t3 = inference.to_tensor(np.ones((3, 4, 5), dtype=np.float64), device="cuda:0")
    assert t3.shape == (3, 1, 1, 4, 5) and t3.dtype == torch.float32    # arr[:, None] twice
    t5 = inference.to_tensor(np.zeros((1, 1, 2, 2, 2), dtype=np.float32), device="cpu")
    assert t5.shape == (1, 1, 2, 2, 2) and t5.device.type == "cpu"


# -------------------------------------------------- any nn.Module on the GPU ---
def test_predict_with_a_generic_module(dev, oracle):
    """predict accepts any module with .parameters() and model(x) -> logits
    (inference.py:155-158): here a plain torch module on the HIP device, so the
    pre/post kernels are checked around torch's own network arithmetic."""
    from aind_exaspim_neuron_segmentation_amd import inference

    torch.manual_seed(3)
    net = torch.nn.Sequential(
        torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.LeakyReLU(0.01), torch.nn.Conv3d(4, 3, 1)
    ).to(dev).eval()
    vol = synthetic.synth_volume((40, 56, 48), seed=5)
    kw = dict(batch_size=3, patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4)

    class CpuTwin:
        """The same weights for the oracle's host loop."""

        def __init__(self, m):
            self.m = torch.nn.Sequential(
                torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.LeakyReLU(0.01), torch.nn.Conv3d(4, 3, 1)
            ).eval()
            self.m.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})

        def __call__(self, x):
            with torch.no_grad():
                return self.m(x)

    want = oracle.predict(vol, CpuTwin(net), **kw)
    got = inference.predict(vol, net, verbose=False, **kw)
    err = np.abs(got - want).max()
    print(f"predict with a generic nn.Module: {err:.3e}")
    assert err < 5e-6
    np.testing.assert_array_equal(got == 0, want == 0)


# ------------------------------------- INTEGRATION.md section 2, verbatim ---
def _integration_stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "class NativeUNet" in b]
    assert len(stub) == 1, "INTEGRATION.md must hold exactly one NativeUNet stub"
    return stub[0]


def test_integration_stub_runs_inside_a_reference_shaped_loop(dev, oracle, golden):
    """Executes the ctypes stub INTEGRATION.md prints (only the library path is
    substituted) and drives it the way the reference's predict does
    (inference.py:85-126: host numpy normalise, per-batch to_tensor -> model ->
    sigmoid -> .cpu() -> trim -> stitch)."""
    from aind_exaspim_neuron_segmentation_amd import _native

    src = _integration_stub_source().replace('"libexaspim_affinity.so"', repr(_native.LIB_PATH))
    ns = {}
    exec(compile(src, "INTEGRATION.md#NativeUNet", "exec"), ns)
    sd = synthetic.synth_state_dict(3, 1, seed=1)
    state = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    model = ns["NativeUNet"](state)

    class AsReferenceModel:
        """What inference.py:155-158 touches: parameters() and __call__."""

        def parameters(self):
            yield torch.zeros(1, device=dev)

        def __call__(self, x):
            return model(x.to(dev))      # to_tensor(..., device) of inference.py:192

    vol = synthetic.synth_volume((160, 160, 160), seed=0)
    got = oracle.predict(vol, AsReferenceModel(), batch_size=8)
    g = golden("g6_default_160.npz")
    err = np.abs(got[:, ::5, ::5, ::5] - g["pred_sub"]).max()
    print(f"INTEGRATION.md stub in the reference-shaped loop vs reference: {err:.3e}")
    assert err < 5e-6


# ------------------------------------------------------------ fp16 range ---
def test_fp16_mode_has_headroom_and_saturates(dev, oracle):
    """The benchmarked 16-bit mode stores activations as IEEE half (max 65504).
    (1) With every input voxel at the clip ceiling (normalised value 1.0) and the
    BatchNorm scales of the first DoubleConv x8 (level-0 activations 64x their
    usual size) nothing overflows and the logits still match the fp32 oracle to
    half precision. (2) With EVERY BatchNorm scale x8 (activations grow 8x per
    layer, 8^18 overall: far outside anything a trained network produces) the
    stores saturate at +-65504: the result stays finite, no inf/nan."""
    from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D

    def model_with(sd):
        m = UNet3D(output_channels=3, compute_dtype="fp16")
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
        return m.to(dev).eval()

    x = torch.ones((2, 1, 32, 32, 32), dtype=torch.float32)
    x[1, 0, ::2] = 0.0                                    # second patch: alternating planes
    sd = synthetic.synth_state_dict(3, 1, seed=1)
    for k in ("inc.double_conv.1.weight", "inc.double_conv.4.weight"):
        sd[k] = sd[k] * 8.0
    want = oracle.unet_forward(x, oracle.OracleModel(sd).sd).numpy()
    got = model_with(sd)(x.to(dev)).cpu().numpy()
    assert np.isfinite(got).all()
    rel = np.abs(got - want).max() / np.abs(want).max()
    print(f"fp16, level-0 activations x64: max|logit| {np.abs(want).max():.1f}, rel. error {rel:.2e}")
    assert rel < 5e-3

    sd_all = synthetic.synth_state_dict(3, 1, seed=1)
    for k in sd_all:
        if k.endswith((".1.weight", ".4.weight")):
            sd_all[k] = sd_all[k] * 8.0
    out = model_with(sd_all).run(x.to(dev), apply_sigmoid=True).cpu().numpy()
    assert np.isfinite(out).all() and out.min() >= 0.0 and out.max() <= 1.0


def test_auto_compute_dtype_checks_the_checkpoint_on_real_patches(dev, oracle, golden, tmp_path):
    """compute_dtype="auto" (load_model accepts it): the first batch predict() gathers runs through
    the float32 and the fp16 engine with a range probe (exaspim_unet_forward_absmax).
    (1) The seeded network passes: fp16 is chosen, the report shows the deviation (< 1e-3) and
        per-layer ranges far below 65504, and predict() equals the explicit fp16 model bit for bit.
    (2) A checkpoint whose BatchNorm scales are all x8 (activations grow 8x per layer) saturates
        half precision: auto falls back to float32 WITH a warning, and its predict() equals the
        float32 model's bit for bit -- where the plain fp16 engine returns something else.
    (3) The probe's ranges are the oracle's: max |activation| of the stored layers, float32 engine."""
    import warnings

    from aind_exaspim_neuron_segmentation_amd import inference
    from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D

    kw = dict(patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4, batch_size=4, verbose=False)
    vol = synthetic.synth_volume((56, 56, 56), seed=5)

    def save(sd, name):
        path = tmp_path / name
        torch.save({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, path)
        return str(path)

    sd = synthetic.synth_state_dict(3, 1, seed=1)
    # (1) passes
    auto = inference.load_model(save(sd, "ok.pth"), compute_dtype="auto")
    assert auto.needs_resolution() and auto.active_dtype() == "fp32"
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = inference.predict(vol, auto, **kw)
    rep = auto.auto_report
    assert auto.active_dtype() == "fp16" and rep["chosen"] == "fp16" and not rep["reasons"]
    assert 0 < rep["max_abs_diff"] < 1e-3 and not rep["saturated"] and rep["fp32_peak"] < 100
    assert len(rep["fp32_absmax"]) == 22 and all(v > 0 for v in rep["fp32_absmax"][:18])
    assert rep["fp32_absmax"][18:] == [0.0] * 4                   # no transposed convolutions here
    fp16 = inference.load_model(save(sd, "ok2.pth"), compute_dtype="fp16")
    np.testing.assert_array_equal(got, inference.predict(vol, fp16, **kw))

    # (3) the probe against the oracle's intermediates (float32 engine, stored layers)
    x = inference._get_batch_inputs(inference.DeviceVolume.from_array(vol, dev),
                                    torch.tensor([[0, 0, 0], [24, 24, 24]], dtype=torch.int32, device=dev),
                                    (32, 32, 32), dev, clip=np.uint16(1000), mn=19.0, mx=1000.0)
    _, feats = oracle.unet_forward(x.cpu(), oracle.OracleModel(sd).sd, return_intermediates=True)
    _, r32 = auto._forward_absmax(x, "fp32")
    r32 = r32.cpu().numpy()
    # conv index (1 + i) of the second conv of each block: x1..x5 = inc.3, down1.3 .. down4.3; y1..y4 = up1.3 .. up4.3
    for name, slot in (("x1", 1), ("x2", 3), ("x3", 5), ("x4", 7), ("x5", 9), ("y1", 11), ("y2", 13), ("y3", 15), ("y4", 17)):
        want = float(feats[name].abs().max())
        assert abs(r32[slot] - want) <= 1e-4 * want, (name, r32[slot], want)

    # (2) a checkpoint that does not fit half precision
    sd_all = synthetic.synth_state_dict(3, 1, seed=1)
    for k in sd_all:
        if k.endswith((".1.weight", ".4.weight")):
            sd_all[k] = sd_all[k] * 8.0
    bad = inference.load_model(save(sd_all, "bad.pth"), compute_dtype="auto")
    with pytest.warns(RuntimeWarning, match="falling back to float32"):
        got_bad = inference.predict(vol, bad, **kw)
    assert bad.active_dtype() == "fp32" and bad.auto_report["saturated"] and bad.auto_report["reasons"]
    f32 = inference.load_model(save(sd_all, "bad2.pth"))
    np.testing.assert_array_equal(got_bad, inference.predict(vol, f32, **kw))
    h16 = inference.load_model(save(sd_all, "bad3.pth"), compute_dtype="fp16")
    assert np.abs(inference.predict(vol, h16, **kw) - got_bad).max() > 1e-3
    # new weights: the decision is taken again
    bad.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        np.testing.assert_array_equal(inference.predict(vol, bad, **kw), got)
    assert bad.active_dtype() == "fp16"
    # the transposed-convolution variant reports its four extra layers
    sdt = synthetic.synth_state_dict(3, 1, seed=2, trilinear=False)
    mt = UNet3D(output_channels=3, trilinear=False, compute_dtype="auto")
    mt.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sdt.items()})
    mt.to(dev).eval()
    rept = mt.fp16_report(x)
    assert all(v > 0 for v in rept["fp32_absmax"][18:]) and rept["max_abs_diff"] < 1e-3
