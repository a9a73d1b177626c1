"""
ctypes binding of the C-ABI HIP extension (include/exaspim_affinity.h).

The product path has no CPU fallback: if the shared library is missing or a
call fails, an exception is raised.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libexaspim_affinity.so")

DT_F32, DT_BF16, DT_F16 = 0, 1, 2
VOX_U8, VOX_U16, VOX_I16, VOX_F32, VOX_F64 = 0, 1, 2, 3, 4
IN_F32, IN_PADDED_F32, IN_PADDED_SPLIT_F16, IN_PADDED_SPLIT_BF16 = 0, 1, 2, 3   # EXASPIM_IN_*
UP_CONVT = 0x100   # EXASPIM_UP_CONVT: OR into a dtype code for UNet3D(trilinear=False)
OPT_SEPARATE_POOL, OPT_SEPARATE_DEEP_POOLS, OPT_PLAIN_UPSAMPLE, OPT_FIRST_PER_GROUP = 1, 2, 4, 8   # EXASPIM_OPT_*
OPT_UPSAMPLE_PER_THREAD = 16

DTYPE_CODES = {
    "fp32": DT_F32, "float32": DT_F32, "f32": DT_F32,
    "bf16": DT_BF16, "bfloat16": DT_BF16,
    "fp16": DT_F16, "float16": DT_F16, "f16": DT_F16,
}


class Block(ctypes.Structure):
    """exaspim_block: local dims, global origin and global shape of a block."""

    _fields_ = [
        ("dims", ctypes.c_int32 * 3),
        ("origin", ctypes.c_int32 * 3),
        ("global_", ctypes.c_int32 * 3),
    ]

    @classmethod
    def make(cls, dims, origin=None, global_shape=None):
        origin = (0, 0, 0) if origin is None else origin
        global_shape = dims if global_shape is None else global_shape
        b = cls()
        b.dims[:] = [int(v) for v in dims]
        b.origin[:] = [int(v) for v in origin]
        b.global_[:] = [int(v) for v in global_shape]
        return b


class Window(ctypes.Structure):
    """exaspim_window: patch shape, overlap and trim of the sliding window."""

    _fields_ = [
        ("patch", ctypes.c_int32 * 3),
        ("overlap", ctypes.c_int32 * 3),
        ("trim", ctypes.c_int32),
    ]

    @classmethod
    def make(cls, patch, overlap, trim):
        w = cls()
        w.patch[:] = [int(v) for v in patch]
        w.overlap[:] = [int(v) for v in overlap]
        w.trim = int(trim)
        return w


_lib = None

_I32x5 = ctypes.c_int32 * 5
_I32x3 = ctypes.c_int32 * 3
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_i32 = ctypes.c_int32

# name -> (restype, argtypes): every symbol include/exaspim_affinity.h declares
SIGNATURES = {
    "exaspim_abi_version": (_i32, []),
    "exaspim_last_error": (ctypes.c_char_p, []),
    "exaspim_unet_param_count": (_sz, [_I32x5, _i32, _i32]),
    "exaspim_unet_packed_bytes": (_sz, [_I32x5, _i32, _i32]),
    "exaspim_unet_pack_weights": (_i32, [_I32x5, _i32, _i32, _vp, _sz, _vp, _sz]),
    "exaspim_unet_create": (_i32, [_I32x5, _i32, _i32, _i32, _vp, _sz, ctypes.POINTER(_vp)]),
    "exaspim_unet_destroy": (None, [_vp]),
    "exaspim_unet_workspace_bytes": (_sz, [_vp, _i32, _i32, _i32, _i32]),
    "exaspim_unet_forward": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "exaspim_unet_forward_trimmed": (
        _i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]
    ),
    "exaspim_unet_input_layout": (_i32, [_vp]),
    "exaspim_unet_forward_prepared": (
        _i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "exaspim_unet_forward_absmax": (
        _i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "exaspim_unet_set_options": (_i32, [_vp, ctypes.c_uint32]),
    "exaspim_unet_timing_begin": (_i32, [_vp, ctypes.c_uint32]),
    "exaspim_unet_timing_read": (_i32, [_vp, ctypes.POINTER(ctypes.c_double * 17),
                                        ctypes.POINTER(ctypes.c_int32 * 17)]),
    "exaspim_histogram": (_i32, [_vp, _i32, _sz, ctypes.c_double, _i32, _i32, ctypes.c_uint32, _vp, _vp]),
    "exaspim_histogram_wide": (_i32, [_vp, _i32, _sz, ctypes.c_double, _i32, _i32, ctypes.c_uint64, _vp, _vp]),
    "exaspim_gather_patches": (_i32, [_vp, _i32, ctypes.POINTER(Block), _vp, _i32, _I32x3,
                                      ctypes.c_double, _i32, ctypes.c_double, ctypes.c_double, _vp, _vp]),
    "exaspim_gather_patches_as": (_i32, [_vp, _i32, ctypes.POINTER(Block), _vp, _i32, _I32x3,
                                         ctypes.c_double, _i32, ctypes.c_double, ctypes.c_double, _i32,
                                         _vp, _vp]),
    "exaspim_stitch_accumulate": (_i32, [_vp, _vp, _i32, _i32, ctypes.POINTER(Window), _vp,
                                         ctypes.POINTER(Block), _vp]),
    "exaspim_stitch_finalize": (_i32, [_vp, _i32, ctypes.POINTER(Window), ctypes.POINTER(Block), _vp]),
    "exaspim_export_f16": (_i32, [_vp, _vp, ctypes.c_size_t, _vp]),
    "exaspim_synth_volume_u16": (_i32, [_vp, ctypes.POINTER(Block), ctypes.c_uint64, _vp]),
}


def lib():
    """
    Loads (once) and returns the shared library with typed signatures.

    Raises
    ------
    RuntimeError
        If the extension has not been built.
    """
    global _lib
    if _lib is None:
        path = LIB_PATH
        # measurement aid (tools/ab_lib.sh): A/B another build of the same ABI without copying it
        # over the product library; never silent
        override = os.environ.get("EXASPIM_LIB")
        if override:
            import sys

            path = os.path.abspath(override)
            print(f"exaspim: EXASPIM_LIB is set, loading {path} instead of the in-tree library",
                  file=sys.stderr)
        if not os.path.exists(path):
            raise RuntimeError(
                f"HIP extension not built: {path} is missing. Build it with "
                "`make -C aind_exaspim_neuron_segmentation_amd/csrc` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). "
                "There is no CPU fallback."
            )
        handle = ctypes.CDLL(path)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def last_error():
    """Returns the library's thread-local error message."""
    msg = lib().exaspim_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what):
    """
    Raises the Python exception matching a C-ABI error code.
    """
    if rc == 0:
        return
    msg = f"{what}: {last_error()} (code {rc})"
    if rc == -1:
        raise ValueError(msg)
    raise RuntimeError(msg)


def channels_array(channels):
    """Converts the five level widths to the int32[5] the ABI takes."""
    if len(channels) != 5:
        raise ValueError("expected five channel widths")
    return _I32x5(*[int(c) for c in channels])


def int3(values):
    """Converts a 3-tuple to int32[3]."""
    return _I32x3(*[int(v) for v in values])
