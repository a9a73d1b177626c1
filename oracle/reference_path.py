"""
ORACLE -- TEST INFRASTRUCTURE ONLY. Never imported by the product package.

CPU restatement of the reference's sliding-window affinity-inference path
(AllenNeuralDynamics/aind-exaspim-neuron-segmentation @ 2025-12-05). Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module, and only as the checker / the timed CPU baseline.

Host arithmetic (clip, percentile normalisation, patch enumeration, reflect
padding, trimmed overlap-add, weight division) is restated in numpy. The
network arithmetic lives in the third-party dependency ``torch`` (unpinned in
the reference, pyproject.toml:30; 2.10.0+rocm7.0 CPU path here), exactly as it
does for the reference, so the U-Net is restated as explicit
``torch.nn.functional`` calls on CPU tensors driven by a plain state_dict --
no ``nn.Module`` of the reference is used or copied.

Parity pin: the reference's own tests pin nothing on this path
(tests/test_example.py:14-17 asserts 1 == 1). This oracle is therefore pinned
against outputs of the reference itself, imported in the build container by
``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``
(checked by ``tests/test_oracle_golden.py``).

Each function cites the reference file:line it follows
(paths relative to src/aind_exaspim_neuron_segmentation/).
"""

import itertools

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # torch.nn.BatchNorm3d default, unet3d.py:144,147
LEAKY_SLOPE = 0.01  # unet3d.py:145,148


# --- helpers: inference.py:340-397 ---
def count_patches(img_shape, patch_shape, overlap):
    """Follows inference.py:340-365."""
    assert len(img_shape) == 5, "Image must have shape (1, 1, D, H, W)"
    n = 1
    for d, ps, ov in zip(img_shape[2:], patch_shape, overlap):
        n *= len(range(0, d - ps + (ps - ov), ps - ov))
    return n


def generate_patch_starts(img_shape, patch_shape, overlap):
    """Follows inference.py:368-397 (z outermost, x fastest)."""
    assert len(img_shape) == 5, "Image must have shape (1, 1, D, H, W)"
    ranges = [
        range(0, d - ps + (ps - ov), ps - ov)
        for d, ps, ov in zip(img_shape[2:], patch_shape, overlap)
    ]
    for start in itertools.product(*ranges):
        yield start


# --- helpers: utils/img_util.py ---
def normalize(img, apply_clip=True, percentiles=(1, 99.9)):
    """Follows utils/img_util.py:504-533 (float64 result)."""
    mn, mx = np.percentile(img, percentiles)
    img = (img - mn) / (mx - mn + 1e-8)
    return np.clip(img, 0, 1) if apply_clip else img


def get_patch_slices(start, patch_shape, img_shape):
    """Follows utils/img_util.py:405-428."""
    return tuple(
        slice(s, min(s + ps, d))
        for s, ps, d in zip(start, patch_shape, img_shape)
    )


def add_padding(patch, patch_shape):
    """Follows utils/img_util.py:362-379 (high-side reflect padding)."""
    pad_width = [(0, ps - s) for ps, s in zip(patch_shape, patch.shape)]
    return np.pad(patch, pad_width, mode="reflect")


def reflect_index(j, n):
    """
    Closed form of numpy's high-side 'reflect' padding: source index inside a
    length-n axis for padded position j >= 0 (period 2(n-1); a length-1 axis
    repeats its only element). Restated so the HIP gather kernel has an
    arithmetic rule to match; checked against np.pad in the tests.
    """
    if n == 1:
        return 0
    m = j % (2 * (n - 1))
    return m if m < n else 2 * (n - 1) - m


# --- network: machine_learning/unet3d.py ---
def _t(sd, key):
    v = sd[key]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def double_conv(x, sd, prefix):
    """Follows DoubleConv, unet3d.py:142-149,165 (BatchNorm in eval mode)."""
    for conv_idx, bn_idx in ((0, 1), (3, 4)):
        x = F.conv3d(
            x,
            _t(sd, f"{prefix}.{conv_idx}.weight"),
            _t(sd, f"{prefix}.{conv_idx}.bias"),
            padding=1,
        )
        x = F.batch_norm(
            x,
            _t(sd, f"{prefix}.{bn_idx}.running_mean"),
            _t(sd, f"{prefix}.{bn_idx}.running_var"),
            _t(sd, f"{prefix}.{bn_idx}.weight"),
            _t(sd, f"{prefix}.{bn_idx}.bias"),
            training=False,
            eps=BN_EPS,
        )
        x = F.leaky_relu(x, LEAKY_SLOPE, inplace=True)   # in place like the reference (unet3d.py:145,148)
    return x


def down(x, sd, name):
    """Follows Down, unet3d.py:194-196,212."""
    return double_conv(
        F.max_pool3d(x, 2), sd, f"{name}.maxpool_conv.1.double_conv"
    )


def up(x1, x2, sd, name):
    """Follows Up, unet3d.py:247-258,280-289 (trilinear upsampling, or
    ConvTranspose3d(k=2, s=2) when the state_dict carries "<name>.up.weight"),
    including the 2-D leftover padding rule (depth is never padded)."""
    if f"{name}.up.weight" in sd:
        x1 = F.conv_transpose3d(
            x1, _t(sd, f"{name}.up.weight"), _t(sd, f"{name}.up.bias"), stride=2
        )
    else:
        x1 = F.interpolate(
            x1, scale_factor=2, mode="trilinear", align_corners=True
        )
    diff_y = x2.size()[2] - x1.size()[2]
    diff_x = x2.size()[3] - x1.size()[3]
    x1 = F.pad(
        x1,
        [diff_x // 2, diff_x - diff_x // 2, diff_y // 2, diff_y - diff_y // 2],
    )
    return double_conv(
        torch.cat([x2, x1], dim=1), sd, f"{name}.conv.double_conv"
    )


def unet_forward(x, sd, return_intermediates=False):
    """Follows UNet3D.forward, unet3d.py:93-105; returns logits."""
    with torch.no_grad():
        x1 = double_conv(x, sd, "inc.double_conv")
        x2 = down(x1, sd, "down1")
        x3 = down(x2, sd, "down2")
        x4 = down(x3, sd, "down3")
        x5 = down(x4, sd, "down4")
        y1 = up(x5, x4, sd, "up1")
        y2 = up(y1, x3, sd, "up2")
        y3 = up(y2, x2, sd, "up3")
        y4 = up(y3, x1, sd, "up4")
        logits = F.conv3d(
            y4, _t(sd, "outc.conv.weight"), _t(sd, "outc.conv.bias")
        )
    if return_intermediates:
        return logits, dict(
            x1=x1, x2=x2, x3=x3, x4=x4, x5=x5, y1=y1, y2=y2, y3=y3, y4=y4
        )
    return logits


class OracleModel:
    """Callable standing in for the reference's eval-mode UNet3D on CPU."""

    def __init__(self, state_dict):
        self.sd = {
            k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v)))
            for k, v in state_dict.items()
        }

    def __call__(self, x):
        return unet_forward(x, self.sd)


# --- prediction: inference.py:29-192 ---
def get_batch_inputs(img, starts, patch_shape):
    """Follows _get_batch_inputs, inference.py:188-192 (float32 CPU tensor)."""
    inputs = np.zeros((len(starts), 1) + tuple(patch_shape), dtype=np.float32)
    for i, start in enumerate(starts):
        s = get_patch_slices(start, patch_shape, img.shape[2:])
        inputs[i, 0, ...] = add_padding(img[(0, 0, *s)], patch_shape)
    return torch.tensor(inputs).to("cpu", dtype=torch.float32)


def predict_batch(img, model, starts, patch_shape, trim=8):
    """Follows _predict_batch, inference.py:155-163."""
    inputs = get_batch_inputs(img, starts, patch_shape)
    with torch.no_grad():
        outputs = torch.sigmoid(model(inputs)).cpu().numpy()
    if trim > 0:
        outputs = outputs[..., trim:-trim, trim:-trim, trim:-trim]
    return outputs


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
):
    """Follows predict, inference.py:79-126 (no progress bar)."""
    img = np.minimum(img, brightness_clip)
    img = normalize(img, percentiles=normalization_percentiles)
    while len(img.shape) < 5:
        img = img[np.newaxis, ...]

    n_patches = count_patches(img.shape, patch_shape, overlap)
    starts_generator = generate_patch_starts(img.shape, patch_shape, overlap)

    n_channels = 3 if affinity_mode else 1
    accum_pred = np.zeros((n_channels,) + img.shape[2:], dtype=np.float32)
    accum_wgt = np.zeros(img.shape[2:], dtype=np.float16)
    for _ in range(0, n_patches, batch_size):
        starts = list(itertools.islice(starts_generator, batch_size))
        patches = predict_batch(img, model, starts, patch_shape, trim=trim)
        for patch, start in zip(patches, starts):
            s = [max(si + trim, 0) for si in start]
            e = [
                min(si + pi, di)
                for si, pi, di in zip(s, patch.shape[1:], img.shape[2:])
            ]
            pred_slices = tuple(slice(si, ei) for si, ei in zip(s, e))
            patch_slices = tuple(slice(0, ei - si) for si, ei in zip(s, e))
            accum_pred[(slice(None),) + pred_slices] += patch[
                (slice(None),) + patch_slices
            ]
            accum_wgt[pred_slices] += 1

    np.divide(accum_pred, accum_wgt, out=accum_pred, where=accum_wgt != 0)
    return accum_pred if affinity_mode else accum_pred[0]
