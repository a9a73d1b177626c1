"""Per-layer HIP-event times of the 17 MFMA convolutions inside real 1024^3-style steps
(run on the MI355X box):  python tools/layer_times.py [--size 512] [--steps 1] [--dtype fp16]
Prints ms per step, and per layer the average launch in us (timing hooks of the C ABI)."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from aind_exaspim_neuron_segmentation_amd import _native, inference  # noqa: E402
from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D  # noqa: E402
from aind_exaspim_neuron_segmentation_amd.utils import synthetic  # noqa: E402

NAMES = ["inc.3", "down1.0", "down1.3", "down2.0", "down2.3", "down3.0", "down3.3", "down4.0", "down4.3",
         "up1.0", "up1.3", "up2.0", "up2.3", "up3.0", "up3.3", "up4.0", "up4.3"]

p = argparse.ArgumentParser()
p.add_argument("--size", type=int, default=512)
p.add_argument("--steps", type=int, default=1)
p.add_argument("--dtype", default="fp16")
p.add_argument("--batch", type=int, default=16)
p.add_argument("--streams", type=int, default=1)
p.add_argument("--mask", type=lambda s: int(s, 0), default=0x1FFFF)
p.add_argument("--options", type=lambda s: int(s, 0), default=0, help="engine option bits (_native.OPT_*)")
a = p.parse_args()

dev = torch.device("cuda:0")
sd = synthetic.synth_state_dict(3, 1, seed=1)
model = UNet3D(output_channels=3, compute_dtype=a.dtype)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model.to(dev).eval()
model.engine_options = a.options
shape = (a.size,) * 3
vol_t = torch.empty(shape, dtype=torch.int16, device=dev)
blk = _native.Block.make(shape, (0, 0, 0), shape)
lib = _native.lib()
_native.check(lib.exaspim_synth_volume_u16(vol_t.data_ptr(), blk, 0, None), "synth")
volume = inference.DeviceVolume(vol_t, np.uint16, (0, 0, 0), shape)
plan = inference.SlidingWindow(shape, (96, 96, 96), (32, 32, 32), 8)


def step():
    return inference.run_sliding_window(volume, model, plan, 3, a.batch, 1000, 19.0, 1000.0,
                                        n_streams=a.streams)


with torch.cuda.device(dev):
    model._ensure_engine(dev)
step()
torch.cuda.synchronize()
_native.check(lib.exaspim_unet_timing_begin(model._engine, a.mask), "timing_begin")
t0 = time.perf_counter()
for _ in range(a.steps):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
ms = (ctypes.c_double * 17)()
cnt = (ctypes.c_int32 * 17)()
_native.check(lib.exaspim_unet_timing_read(model._engine, ctypes.byref(ms), ctypes.byref(cnt)), "timing_read")
nb = -(-len(plan.starts()) // a.batch)
print(f"ms_per_step {dt * 1e3:.1f}  ({nb} batches, {dt * 1e6 / nb:.0f} us per batch)")
tot = 0.0
line = []
for i, n in enumerate(NAMES):
    if cnt[i]:
        us = ms[i] * 1e3 / cnt[i]
        tot += us
        line.append(f"{n} {us:.0f}")
print("us per launch: " + "  ".join(line) + f"  | sum {tot:.0f}")
