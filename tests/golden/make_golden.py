"""
Generates the golden fixtures under tests/golden/ by importing the REFERENCE
implementation (read-only mount at /root/reference) in the build container.

Run once, here:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference never travels to the GPU box; only the .npz data written by this
script does. Eight third-party imports that the reference makes at module top
but never touches on the predict()/load_model() path (kimimaro, waterz,
fastremap, gcsfs, s3fs, tifffile, zarr, google.cloud.storage) are absent from
this image and are replaced by empty stub modules.

Inputs are produced by aind_exaspim_neuron_segmentation_amd.utils.synthetic
(pure functions of a seed), so tests regenerate them instead of storing them.
"""

import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

for _name in [
    "kimimaro", "waterz", "gcsfs", "s3fs", "tifffile", "zarr",
    "google", "google.cloud", "google.cloud.storage",
]:
    sys.modules[_name] = types.ModuleType(_name)
_fr = types.ModuleType("fastremap")
for _n in ("mask_except", "renumber", "unique"):
    setattr(_fr, _n, None)
sys.modules["fastremap"] = _fr
sys.path.insert(0, "/root/reference/src")

import torch  # noqa: E402

from aind_exaspim_neuron_segmentation import inference as ref_inf  # noqa: E402
from aind_exaspim_neuron_segmentation.machine_learning.unet3d import (  # noqa: E402
    UNet3D as RefUNet3D,
)
from aind_exaspim_neuron_segmentation.utils import img_util as ref_img  # noqa: E402

from aind_exaspim_neuron_segmentation_amd.utils import synthetic  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def ref_model_from_synth(output_channels, width_multiplier, seed, via_file=True):
    """Loads synthetic weights through the reference's own load path."""
    sd = synthetic.synth_state_dict(output_channels, width_multiplier, seed)
    tsd = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    if via_file and width_multiplier == 1:
        # exercise inference.load_model (inference.py:400-424) itself
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "synthetic.pth")
            torch.save(tsd, path)
            model = ref_inf.load_model(
                path, affinity_mode=(output_channels == 3), device="cpu"
            )
    else:
        model = RefUNet3D(
            output_channels=output_channels, width_multiplier=width_multiplier
        )
        model.load_state_dict(tsd)
        model.eval()
    return model


def g1_patch_starts():
    cases = [
        # (D, H, W), patch, overlap
        ((96, 96, 96), (96, 96, 96), (32, 32, 32)),
        ((160, 160, 160), (96, 96, 96), (32, 32, 32)),
        ((200, 130, 97), (96, 96, 96), (32, 32, 32)),
        ((32, 96, 96), (96, 96, 96), (32, 32, 32)),      # d <= overlap -> 0
        ((33, 40, 500), (96, 96, 96), (32, 32, 32)),     # d < patch
        ((224, 224, 224), (96, 96, 96), (32, 32, 32)),
        ((56, 40, 48), (32, 32, 32), (8, 8, 8)),
        ((80, 64, 72), (32, 48, 64), (16, 8, 0)),
        ((512, 512, 512), (96, 96, 96), (32, 32, 32)),
    ]
    out = {}
    for i, (vol, ps, ov) in enumerate(cases):
        shape5 = (1, 1) + vol
        starts = np.array(
            list(ref_inf.generate_patch_starts(shape5, ps, ov)), dtype=np.int64
        ).reshape(-1, 3)
        if vol == (512, 512, 512):
            starts = starts[[0, 1, 7, 8, 63, 64, 510, 511]]
        out[f"case{i}_vol"] = np.array(vol)
        out[f"case{i}_patch"] = np.array(ps)
        out[f"case{i}_overlap"] = np.array(ov)
        out[f"case{i}_count"] = np.array(ref_inf.count_patches(shape5, ps, ov))
        out[f"case{i}_starts"] = starts
    out["n_cases"] = np.array(len(cases))
    save("g1_patch_starts.npz", **out)


def g2_normalize():
    out = {}
    vol = synthetic.synth_volume((40, 48, 56), seed=3)
    cases = {
        "u16_clip1000": np.minimum(vol, 1000),
        "u16_noclip": vol,
        "u16_sparse": np.where(vol > 1990, vol * 20, vol // 50).astype(np.uint16),
        "f32": (vol.astype(np.float32) * 0.37 - 50.0),
        "u8": (vol % 251).astype(np.uint8),
        "i16": (vol.astype(np.int32) - 1000).astype(np.int16),
        "const": np.full((8, 8, 8), 7, dtype=np.uint16),
    }
    for name, arr in cases.items():
        for pct_name, pct in (("default", (1, 99.9)), ("alt", (0.5, 75.25))):
            mn, mx = np.percentile(arr, pct)
            res = ref_img.normalize(arr, percentiles=pct)
            assert res.dtype == np.float64
            out[f"{name}_{pct_name}_mnmx"] = np.array([mn, mx], dtype=np.float64)
            out[f"{name}_{pct_name}_out"] = res[::3, ::5, ::7].copy()
    # np.minimum dtype semantics used by predict (inference.py:79)
    out["minimum_u16_dtype"] = np.array(str(np.minimum(vol, 1000).dtype))
    out["minimum_f32_dtype"] = np.array(
        str(np.minimum(vol.astype(np.float32), 1000).dtype)
    )
    save("g2_normalize.npz", **out)


def g2b_padding():
    out = {}
    base = synthetic.synth_volume((33, 20, 1), seed=5).astype(np.float64)
    cases = [
        ((33, 20, 1), (96, 32, 4)),   # multi-reflection + singleton axis
        ((17, 20, 1), (32, 20, 1)),
        ((2, 3, 1), (9, 9, 3)),
    ]
    for i, (sub, ps) in enumerate(cases):
        patch = base[: sub[0], : sub[1], : sub[2]]
        out[f"case{i}_in"] = patch
        out[f"case{i}_patch_shape"] = np.array(ps)
        out[f"case{i}_out"] = ref_img.add_padding(patch, ps)
    out["n_cases"] = np.array(len(cases))
    # get_patch_slices
    sl = ref_img.get_patch_slices((64, 0, 128), (96, 96, 96), (130, 50, 224))
    out["slices"] = np.array([[s.start, s.stop] for s in sl])
    save("g2b_padding.npz", **out)


def g3_tiny_predict():
    """Tiny-width model, ragged volume, full stitched output: pins the host
    pipeline (clip, normalise, gather, reflect, trim, stitch, divide)."""
    model = ref_model_from_synth(3, 0.125, seed=2, via_file=False)
    vol = synthetic.synth_volume((56, 40, 48), seed=7)
    kw = dict(
        batch_size=3, patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4,
        verbose=False,
    )
    pred = ref_inf.predict(vol, model, **kw)
    assert pred.dtype == np.float32 and pred.shape == (3, 56, 40, 48)
    model1 = ref_model_from_synth(1, 0.125, seed=2, via_file=False)
    pred1 = ref_inf.predict(vol, model1, affinity_mode=False, **kw)
    assert pred1.shape == (56, 40, 48)
    # float32 input + no trim + 5-D input
    volf = (vol.astype(np.float32) * 0.5)[None, None]
    kw0 = dict(kw, trim=0, brightness_clip=400, normalization_percentiles=(5, 95))
    pred0 = ref_inf.predict(volf, model, **kw0)
    save(
        "g3_tiny_predict.npz",
        pred=pred, pred_fg=pred1, pred_f32_notrim=pred0[:, ::2, ::2, ::2].copy(),
    )


def g4_single_patch():
    """Full-width model on one 96^3 patch: logits, sigmoid and per-stage
    checksums (pins BN folding, layer order, skip-cat order, upsampling)."""
    model = ref_model_from_synth(3, 1, seed=1)
    vol = synthetic.synth_volume((96, 96, 96), seed=0)
    img = ref_img.normalize(np.minimum(vol, 1000))
    x = torch.tensor(img[None, None].astype(np.float32))
    feats = {}

    def hook(name):
        def fn(_m, _i, o):
            feats[name] = o.detach()
        return fn

    names = ["inc", "down1", "down2", "down3", "down4", "up1", "up2", "up3", "up4"]
    handles = [getattr(model, n).register_forward_hook(hook(n)) for n in names]
    with torch.no_grad():
        logits = model(x)
    for h in handles:
        h.remove()
    out = dict(
        logits_sub=logits[0, :, ::8, ::8, ::8].numpy().copy(),
        logits_slab=logits[0, :, 40:44, 17:21, :].numpy().copy(),
        sigmoid_sub=torch.sigmoid(logits)[0, :, ::8, ::8, ::8].numpy().copy(),
        logits_stats=np.array(
            [logits.mean().item(), logits.abs().mean().item(),
             logits.min().item(), logits.max().item()], dtype=np.float64),
    )
    for n in names:
        f = feats[n][0].double()
        out[f"{n}_stats"] = np.array(
            [f.mean().item(), f.abs().mean().item(), f.min().item(), f.max().item()]
        )
        step = max(1, f.shape[1] // 6)
        out[f"{n}_sub"] = feats[n][0, ::4, ::step, ::step, ::step].numpy().copy()
    # predict() on the same single patch (inference.py:29-126)
    pred = ref_inf.predict(vol, model, verbose=False)
    out["predict_sub"] = pred[:, ::4, ::4, ::4].copy()
    out["predict_nonzero_bbox"] = np.array(
        [[np.nonzero(pred.any(axis=(0, 2, 3)) if a == 0 else
                     pred.any(axis=(0, 1, 3)) if a == 1 else
                     pred.any(axis=(0, 1, 2)))[0][[0, -1]]] for a in range(3)]
    ).reshape(3, 2)
    save("g4_single_patch.npz", **out)


def g5_fullwidth_small_patches():
    """Full-width model, 32^3 patches, ragged volume: the case the GPU parity
    tests replay end to end at a size the CPU finishes in seconds."""
    model = ref_model_from_synth(3, 1, seed=1)
    vol = synthetic.synth_volume((72, 40, 56), seed=11)
    pred = ref_inf.predict(
        vol, model, batch_size=4, patch_shape=(32, 32, 32), overlap=(8, 8, 8),
        trim=4, verbose=False,
    )
    model1 = ref_model_from_synth(1, 1, seed=4)
    pred1 = ref_inf.predict(
        vol, model1, affinity_mode=False, batch_size=5, patch_shape=(32, 32, 32),
        overlap=(16, 16, 16), trim=2, verbose=False,
    )
    save("g5_fullwidth_small.npz", pred=pred[:, ::2, ::2, ::2].copy(),
         pred_row=pred[:, 30, 20, :].copy(), pred_fg=pred1[::2, ::2, ::2].copy())


def g6_default_config():
    """Reference defaults (96^3 patches, overlap 32, trim 8) on 160^3 (exact
    fit: top 8 voxels stay zero) -- subsampled output."""
    model = ref_model_from_synth(3, 1, seed=1)
    vol = synthetic.synth_volume((160, 160, 160), seed=0)
    pred = ref_inf.predict(vol, model, batch_size=8, verbose=False)
    zero = (pred == 0).all(axis=0)
    save(
        "g6_default_160.npz",
        pred_sub=pred[:, ::5, ::5, ::5].copy(),
        pred_line=pred[:, 80, 81, :].copy(),
        zero_fraction=np.array(zero.mean()),
        zero_z=zero.all(axis=(1, 2)), zero_y=zero.all(axis=(0, 2)),
        zero_x=zero.all(axis=(0, 1)),
    )


def g7_conv_transpose_variant():
    """UNet3D(trilinear=False): ConvTranspose3d(k=2, s=2) in the Up blocks
    (unet3d.py:254-258) -- full-width forward on 32^3 patches and a small predict."""
    sd = synthetic.synth_state_dict(3, 1, seed=8, trilinear=False)
    model = RefUNet3D(output_channels=3, trilinear=False)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    model.eval()
    assert len(sd) == 136
    vols = [synthetic.synth_volume((32, 32, 48), seed=60 + i) for i in range(2)]
    x = np.stack([ref_img.normalize(np.minimum(v, 1000)) for v in vols])[:, None]
    with torch.no_grad():
        logits = model(torch.tensor(x.astype(np.float32)))
    vol = synthetic.synth_volume((56, 40, 48), seed=61)
    pred = ref_inf.predict(vol, model, batch_size=3, patch_shape=(32, 32, 32),
                           overlap=(8, 8, 8), trim=4, verbose=False)
    save("g7_conv_transpose.npz", logits_sub=logits[:, :, ::2, ::2, ::2].numpy().copy(),
         logits_row=logits[1, :, 17, 9, :].numpy().copy(),
         pred_sub=pred[:, ::2, ::2, ::2].copy())


def g8_validation_tiling():
    """The deterministic validation tiling of the reference's trainer
    (data_handling.py:402-413: generate_patch_starts with zero overlap, patch
    centres, img_util.is_contained with a 64-voxel buffer), produced with the
    reference's own two helpers."""
    out = {}
    cases = [((500, 420, 640), (96, 96, 96)), ((192, 192, 192), (96, 96, 96)),
             ((330, 300, 270), (64, 64, 64))]
    for i, (vol, ps) in enumerate(cases):
        shape5 = (1, 1) + vol
        starts = list(ref_inf.generate_patch_starts(shape5, ps, (0, 0, 0)))
        centers = [[v + s // 2 for v, s in zip(st, ps)] for st in starts]
        kept = [c for c in centers if ref_img.is_contained(c, vol, buffer=64)]
        out[f"case{i}_vol"] = np.array(vol)
        out[f"case{i}_patch"] = np.array(ps)
        out[f"case{i}_starts"] = np.array(starts, dtype=np.int64).reshape(-1, 3)
        out[f"case{i}_centers_kept"] = np.array(kept, dtype=np.int64).reshape(-1, 3)
    out["n_cases"] = np.array(len(cases))
    save("g8_validation_tiling.npz", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["g1", "g2", "g2b", "g3", "g4", "g5", "g6", "g7", "g8"]
    table = dict(g1=g1_patch_starts, g2=g2_normalize, g2b=g2b_padding,
                 g3=g3_tiny_predict, g4=g4_single_patch,
                 g5=g5_fullwidth_small_patches, g6=g6_default_config,
                 g7=g7_conv_transpose_variant, g8=g8_validation_tiling)
    for w in which:
        table[w]()
