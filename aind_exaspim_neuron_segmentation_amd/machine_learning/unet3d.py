"""
MI355X-native stand-in for the reference's ``UNet3D``
(machine_learning/unet3d.py:16-105 of the reference).

The module tree below exists only to own parameters under the reference's
state_dict keys (``inc.double_conv.0.weight`` ...), so checkpoints written by
the reference's trainer (train.py:286) load unchanged with
``load_state_dict(strict=True)``. No torch operator runs in ``forward``: the
parameters are BatchNorm-folded and packed once by the C-ABI extension, and
the whole network (3x3x3 convs on the matrix cores, max-pool, trilinear
upsampling or 2x2x2 transposed convolution, skip concatenation, 1x1x1 head) executes as hand-written gfx950
kernels on the caller's HIP stream (csrc/engine.hip).

Inference only (eval-mode BatchNorm; both ``trilinear`` variants); there is no CPU
path -- calling the model with a CPU tensor raises.
"""

import ctypes
import math
import warnings

import numpy as np
import torch
from torch import nn

from aind_exaspim_neuron_segmentation_amd import _native
from aind_exaspim_neuron_segmentation_amd.machine_learning.spec import (
    unet_channels,
    unet_layer_specs,
)


class _Holder(nn.Module):
    """Anonymous node of the parameter tree (no computation of its own)."""


class _ConvParams(nn.Module):
    """Weight and bias of one convolution, initialised like nn.Conv3d."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        bound = 1.0 / math.sqrt(cin * k ** 3)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)


class _ConvTransposeParams(nn.Module):
    """Weight (in, out, k, k, k) and bias of one nn.ConvTranspose3d."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, k, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        bound = 1.0 / math.sqrt(cout * k ** 3)   # torch's fan_in of a transposed conv
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)


class _NormParams(nn.Module):
    """Affine parameters and running statistics of one BatchNorm3d."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


def _attach(root, dotted, module):
    """Registers "module" under a dotted path, creating holders on the way."""
    node = root
    parts = dotted.split(".")
    for name in parts[:-1]:
        if name not in node._modules:
            node.add_module(name, _Holder())
        node = node._modules[name]
    node.add_module(parts[-1], module)


class UNet3D(nn.Module):
    """
    3D U-Net whose forward pass runs on the exaspim_affinity HIP extension.

    Parameters
    ----------
    output_channels : int, optional
        Number of channels in the output. Default is 1.
    trilinear : bool, optional
        True (default; the variant load_model of the reference creates) for
        trilinear upsampling, False for ConvTranspose3d(k=2, s=2) up blocks.
    width_multiplier : float, optional
        Factor that scales the number of channels in each layer. Default is 1.
    compute_dtype : str, optional
        "fp32" (exact fp32 matrix-core path, default), "bf16" or "fp16"
        (16-bit activations and weights, fp32 accumulation), or "auto": fp16
        if THIS checkpoint stays inside half range and within 1e-3 of its own
        float32 result on the first batch of real patches it sees, float32
        otherwise (resolve_compute_dtype; until then the model runs float32).
    """

    AUTO_TOLERANCE = 1e-3    # north_star's bar on the probabilities
    AUTO_HEADROOM = 2.0      # float32 activations must stay below 65504 / headroom
    HALF_MAX = 65504.0

    def __init__(self, output_channels=1, trilinear=True, width_multiplier=1,
                 compute_dtype="fp32"):
        super().__init__()
        if compute_dtype != "auto" and compute_dtype not in _native.DTYPE_CODES:
            raise ValueError(f"unknown compute_dtype {compute_dtype!r}")
        self.channels = unet_channels(width_multiplier)
        self.trilinear = trilinear
        self.output_channels = output_channels
        self.compute_dtype = compute_dtype
        layers, (head_in, head_out) = unet_layer_specs(
            output_channels, trilinear, width_multiplier
        )
        for kind, prefix, *widths in layers:
            if kind == "conv_transpose":
                cin, cout = widths
                _attach(self, prefix, _ConvTransposeParams(cin, cout, 2))
                continue
            cin, cmid, cout = widths
            _attach(self, f"{prefix}.0", _ConvParams(cin, cmid, 3))
            _attach(self, f"{prefix}.1", _NormParams(cmid))
            _attach(self, f"{prefix}.3", _ConvParams(cmid, cout, 3))
            _attach(self, f"{prefix}.4", _NormParams(cout))
        _attach(self, "outc.conv", _ConvParams(head_in, head_out, 1))
        self._engines = {}        # compute dtype -> {"handle": exaspim_unet*, "packed": device image, "key": state}
        self._resolved = None     # what "auto" decided ("fp16" / "fp32"), None until the first batch
        self._resolved_key = None # parameter state that decision was taken for
        self.auto_report = None   # the measurements behind that decision (resolve_compute_dtype)
        self._workspace = None
        self.engine_options = 0   # _native.OPT_* bits handed to every engine (bit-identical plans: tests, tools)

    # ---- engine management -------------------------------------------------
    def _canonical_params(self):
        """float32 vector in the order exaspim_unet_param_count documents."""
        chunks = []
        for key, value in self.state_dict().items():
            if key.endswith("num_batches_tracked"):
                continue
            chunks.append(value.detach().to("cpu", torch.float32).reshape(-1))
        return torch.cat(chunks).contiguous().numpy()

    def _state_key(self, device, dtype):
        tensors = list(self.parameters()) + list(self.buffers())
        return (
            str(device), dtype,
            tuple((t.data_ptr(), t._version) for t in tensors),
        )

    def active_dtype(self):
        """The compute dtype forward passes run in right now ("auto": float32 until resolved)."""
        if self.compute_dtype == "auto":
            return self._resolved or "fp32"
        return self.compute_dtype

    @property
    def _engine(self):
        """Opaque exaspim_unet* of the active compute dtype (None before the first use)."""
        entry = self._engines.get(self.active_dtype())
        return entry["handle"] if entry else None

    def _release_engine(self):
        for entry in self._engines.values():
            _native.lib().exaspim_unet_destroy(entry["handle"])
        self._engines = {}

    def __del__(self):
        try:
            self._release_engine()
        except Exception:
            pass

    def _ensure_engine(self, device, dtype=None):
        """Packs the current parameters for "dtype" (default: the active one) and binds an engine
        to them; a no-op while neither the parameters nor the device changed."""
        dtype = dtype or self.active_dtype()
        key = self._state_key(device, dtype)
        entry = self._engines.get(dtype)
        if entry is not None and key == entry["key"]:
            return self._apply_options(dtype)
        if entry is not None:
            _native.lib().exaspim_unet_destroy(entry["handle"])
            del self._engines[dtype]
        lib = _native.lib()
        ch = _native.channels_array(self.channels)
        code = _native.DTYPE_CODES[dtype]
        if not self.trilinear:
            code |= _native.UP_CONVT
        params = self._canonical_params()
        expected = lib.exaspim_unet_param_count(ch, self.output_channels, code)
        if expected == 0:
            raise ValueError(f"unsupported network: {_native.last_error()}")
        if params.size != expected:
            raise RuntimeError(
                f"parameter vector has {params.size} values, extension expects {expected}"
            )
        nbytes = lib.exaspim_unet_packed_bytes(ch, self.output_channels, code)
        packed = np.empty(nbytes, dtype=np.uint8)
        _native.check(
            lib.exaspim_unet_pack_weights(
                ch, self.output_channels, code, params.ctypes.data, params.size,
                packed.ctypes.data, nbytes,
            ),
            "exaspim_unet_pack_weights",
        )
        packed_dev = torch.from_numpy(packed).to(device)
        handle = ctypes.c_void_p()
        index = device.index if device.index is not None else torch.cuda.current_device()
        _native.check(
            lib.exaspim_unet_create(
                ch, self.output_channels, code, index, packed_dev.data_ptr(), nbytes,
                ctypes.byref(handle),
            ),
            "exaspim_unet_create",
        )
        self._engines[dtype] = {"handle": handle, "packed": packed_dev, "key": key, "options": 0}
        return self._apply_options(dtype)

    def _apply_options(self, dtype):
        entry = self._engines[dtype]
        if entry["options"] != self.engine_options:
            _native.check(_native.lib().exaspim_unet_set_options(entry["handle"], int(self.engine_options)),
                          "exaspim_unet_set_options")
            entry["options"] = self.engine_options
        return entry["handle"]

    def _get_workspace(self, n, d, h, w, device, stream):
        """Scratch buffer for one forward; one per HIP stream so that batches in
        flight on different streams never share activations."""
        need = _native.lib().exaspim_unet_workspace_bytes(self._engine, n, d, h, w)
        if need == 0:
            raise RuntimeError(
                "Sizes of tensors must match: " + _native.last_error()
            )
        if self._workspace is None:
            self._workspace = {}
        key = (str(device), stream)
        ws = self._workspace.get(key)
        if ws is None or ws.numel() < need:
            self._workspace.pop(key, None)
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._workspace[key] = ws
        return ws

    # ---- 16-bit safety: ranges and deviation on real patches -----------------
    def _forward_absmax(self, x, dtype):
        """sigmoid(forward(x)) with the engine of "dtype" plus the largest |activation| every
        layer stored (exaspim_unet_forward_absmax)."""
        device = x.device
        n, _, d, h, w = x.shape
        with torch.cuda.device(device):
            handle = self._ensure_engine(device, dtype)
            stream = torch.cuda.current_stream(device).cuda_stream
            need = _native.lib().exaspim_unet_workspace_bytes(handle, n, d, h, w)
            if need == 0:
                raise RuntimeError("Sizes of tensors must match: " + _native.last_error())
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            out = torch.empty((n, self.output_channels, d, h, w), dtype=torch.float32, device=device)
            absmax = torch.zeros(22, dtype=torch.float32, device=device)   # EXASPIM_ABSMAX_SLOTS
            _native.check(
                _native.lib().exaspim_unet_forward_absmax(
                    handle, x.data_ptr(), out.data_ptr(), n, d, h, w, 1, absmax.data_ptr(),
                    ws.data_ptr(), ws.numel(), stream),
                "exaspim_unet_forward_absmax",
            )
        return out, absmax

    def fp16_report(self, x):
        """
        Measures, on a batch of real patches, what IEEE-half storage does to THIS checkpoint:
        the float32 engine's largest |activation| per layer (half stores saturate at 65504), the
        fp16 engine's (65504 there = a store saturated) and the largest deviation of the fp16
        probabilities from the float32 ones.

        Parameters
        ----------
        x : torch.Tensor
            float32 (B, 1, D, H, W) network inputs on the HIP device (normalised patches, as
            inference._get_batch_inputs builds them).

        Returns
        -------
        dict
            "max_abs_diff", "fp32_absmax" / "fp16_absmax" (per layer: inc.0, the 17 MFMA
            convolutions, the 4 transposed convolutions), "fp32_peak", "saturated", "finite".
        """
        if x.dim() != 5 or x.shape[1] != 1 or not x.is_cuda:
            raise RuntimeError(f"expected a (B, 1, D, H, W) tensor on a HIP device, got {tuple(x.shape)} on {x.device}")
        x = x.to(torch.float32).contiguous()
        p32, r32 = self._forward_absmax(x, "fp32")
        p16, r16 = self._forward_absmax(x, "fp16")
        diff = float((p16 - p32).abs().max().item())
        r32, r16 = r32.cpu().numpy(), r16.cpu().numpy()
        return {
            "max_abs_diff": diff,
            "fp32_absmax": [float(v) for v in r32],
            "fp16_absmax": [float(v) for v in r16],
            "fp32_peak": float(np.nanmax(r32)) if np.isfinite(r32).any() else float("nan"),
            "saturated": bool((r16 >= self.HALF_MAX).any()),
            "finite": bool(np.isfinite(r32).all() and np.isfinite(r16).all() and np.isfinite(diff)),
            "patches": int(x.shape[0]),
        }

    def needs_resolution(self):
        """True for compute_dtype="auto" before resolve_compute_dtype has run on the CURRENT
        parameters (loading new weights or moving the model asks for a new decision)."""
        if self.compute_dtype != "auto":
            return False
        if self._resolved is not None and self._resolved_key != self._state_key(
                next(self.parameters()).device, "auto"):
            self._resolved, self.auto_report = None, None
        return self._resolved is None

    def resolve_compute_dtype(self, x):
        """
        compute_dtype="auto": decides between fp16 and float32 from fp16_report(x) on the first
        batch of real patches (predict() calls this) -- fp16 if nothing saturated, the float32
        activations keep a factor AUTO_HEADROOM below 65504 and the probabilities agree within
        AUTO_TOLERANCE; float32, with a warning, otherwise. The reference runs any checkpoint in
        float32 (inference.py:400-424); its trainer uses fp16 autocast (train.py:218-223), so
        trained checkpoints normally pass.

        Returns
        -------
        str
            "fp16" or "fp32" (also kept in active_dtype(); the report in auto_report).
        """
        if self.compute_dtype != "auto":
            return self.compute_dtype
        rep = self.fp16_report(x)
        reasons = []
        if not rep["finite"]:
            reasons.append("non-finite activations")
        if rep["saturated"]:
            reasons.append("a half-precision store saturated at 65504")
        if rep["fp32_peak"] * self.AUTO_HEADROOM > self.HALF_MAX:
            reasons.append(f"float32 activations reach {rep['fp32_peak']:.4g} (less than a factor "
                           f"{self.AUTO_HEADROOM:g} below 65504)")
        if not rep["max_abs_diff"] <= self.AUTO_TOLERANCE:
            reasons.append(f"fp16 probabilities deviate by {rep['max_abs_diff']:.3g} > {self.AUTO_TOLERANCE:g}")
        self._resolved = "fp32" if reasons else "fp16"
        self._resolved_key = self._state_key(x.device, "auto")
        rep["chosen"], rep["reasons"] = self._resolved, reasons
        self.auto_report = rep
        if reasons:
            warnings.warn("UNet3D(compute_dtype='auto'): falling back to float32 -- " + "; ".join(reasons),
                          RuntimeWarning, stacklevel=2)
        return self._resolved

    # ---- forward -----------------------------------------------------------
    def run(self, x, apply_sigmoid=False, out=None, trim=0):
        """
        Runs the network on a float32 device tensor of shape (B, 1, D, H, W).

        Parameters
        ----------
        x : torch.Tensor
            Input patches on a HIP device.
        apply_sigmoid : bool, optional
            Fuse the sigmoid of inference.py:158 into the head. Default False.
        out : torch.Tensor, optional
            Preallocated (B, C, D, H, W) float32 output.
        trim : int, optional
            The caller discards the outputs within "trim" voxels of every patch
            face (inference.py:161-162), so they are not computed and those
            voxels of the result are undefined. Default is 0 (everything).

        Returns
        -------
        torch.Tensor
            Logits (or probabilities) with shape (B, output_channels, D, H, W).
        """
        if self.training:
            raise RuntimeError(
                "UNet3D (MI355X) implements eval-mode inference only; call .eval()"
            )
        if not x.is_cuda:
            raise RuntimeError(
                "UNet3D (MI355X) has no CPU path: move the input to a HIP device"
            )
        if x.dim() != 5 or x.shape[1] != 1:
            raise RuntimeError(f"expected input of shape (B, 1, D, H, W), got {tuple(x.shape)}")
        x = x.to(torch.float32).contiguous()
        n, _, d, h, w = x.shape
        if d % 16 or h % 16 or w % 16:
            raise RuntimeError(
                "Sizes of tensors must match: patch dimensions must be multiples of 16, "
                f"got {(d, h, w)}"
            )
        device = x.device
        with torch.cuda.device(device):
            self._ensure_engine(device)
            stream = torch.cuda.current_stream(device).cuda_stream
            ws = self._get_workspace(n, d, h, w, device, stream)
            if out is None:
                out = torch.empty(
                    (n, self.output_channels, d, h, w), dtype=torch.float32, device=device
                )
            _native.check(
                _native.lib().exaspim_unet_forward_trimmed(
                    self._engine, x.data_ptr(), out.data_ptr(), n, d, h, w,
                    1 if apply_sigmoid else 0, int(trim), ws.data_ptr(), ws.numel(), stream,
                ),
                "exaspim_unet_forward_trimmed",
            )
        return out

    def input_layout(self, device=None):
        """
        Layout code (_native.IN_PADDED_*) of the input run_prepared() takes: the
        first convolution's own operand layout for this compute dtype.
        """
        device = next(self.parameters()).device if device is None else device
        with torch.cuda.device(device):
            self._ensure_engine(device)
        code = _native.lib().exaspim_unet_input_layout(self._engine)
        if code < 0:
            raise RuntimeError("exaspim_unet_input_layout failed")
        return int(code)

    def run_prepared(self, prepared, shape, apply_sigmoid=False, out=None, trim=0):
        """
        run() from a batch the gather kernel has already written in the first
        convolution's operand layout (inference._get_batch_inputs(..., layout=
        self.input_layout())): a contiguous 4-byte-per-voxel device tensor of shape
        (B, D + 2, H + 2, W + 2). Same bits as run() on the float32 patches.

        Parameters
        ----------
        prepared : torch.Tensor
            The prepared batch.
        shape : Tuple[int]
            (B, D, H, W) of the patches.
        """
        if self.training:
            raise RuntimeError(
                "UNet3D (MI355X) implements eval-mode inference only; call .eval()"
            )
        n, d, h, w = (int(v) for v in shape)
        if (not prepared.is_cuda or not prepared.is_contiguous() or prepared.element_size() != 4
                or tuple(prepared.shape) != (n, d + 2, h + 2, w + 2)):
            raise RuntimeError(
                f"prepared input must be a contiguous 4-byte device tensor of shape {(n, d + 2, h + 2, w + 2)}"
            )
        if d % 16 or h % 16 or w % 16:
            raise RuntimeError(
                "Sizes of tensors must match: patch dimensions must be multiples of 16, "
                f"got {(d, h, w)}"
            )
        device = prepared.device
        with torch.cuda.device(device):
            self._ensure_engine(device)
            stream = torch.cuda.current_stream(device).cuda_stream
            ws = self._get_workspace(n, d, h, w, device, stream)
            if out is None:
                out = torch.empty(
                    (n, self.output_channels, d, h, w), dtype=torch.float32, device=device
                )
            _native.check(
                _native.lib().exaspim_unet_forward_prepared(
                    self._engine, prepared.data_ptr(), out.data_ptr(), n, d, h, w,
                    1 if apply_sigmoid else 0, int(trim), ws.data_ptr(), ws.numel(), stream,
                ),
                "exaspim_unet_forward_prepared",
            )
        return out

    def forward(self, x):
        """
        Forward pass: (B, 1, D, H, W) -> logits (B, output_channels, D, H, W).
        """
        return self.run(x, apply_sigmoid=False)
