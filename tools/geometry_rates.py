"""Throughput of predict() over patch geometries other than the default (MI355X box):
python tools/geometry_rates.py    -- voxels/s and reference-equivalent TFLOP/s per geometry."""
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

FLOP_96 = 370_145_230_848
dev = torch.device("cuda:0")
sd = synthetic.synth_state_dict(3, 1, seed=1)
model = UNet3D(output_channels=3, compute_dtype="fp16")
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model.to(dev).eval()
size = 768
shape = (size,) * 3
vol_t = torch.empty(shape, dtype=torch.int16, device=dev)
blk = _native.Block.make(shape, (0, 0, 0), shape)
_native.check(_native.lib().exaspim_synth_volume_u16(vol_t.data_ptr(), blk, 0, None), "synth")
volume = inference.DeviceVolume(vol_t, np.uint16, (0, 0, 0), shape)
for patch, overlap, trim, batch in [(96, 32, 8, 16), (128, 32, 8, 8), (64, 16, 4, 32), (160, 32, 8, 4),
                                    (48, 16, 4, 64), (112, 32, 8, 8), (80, 32, 8, 16)]:
    plan = inference.SlidingWindow(shape, (patch,) * 3, (overlap,) * 3, trim)
    n = len(plan.starts())

    def step():
        return inference.run_sliding_window(volume, model, plan, 3, batch, 1000, 19.0, 1000.0)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flop = n * FLOP_96 * (patch / 96.0) ** 3
    print(f"patch {patch:3d} overlap {overlap} trim {trim} batch {batch:2d}: {n:5d} patches, {dt * 1e3:7.1f} ms, "
          f"{size ** 3 / dt / 1e6:7.1f} Mvox/s, {flop / dt / 1e12:6.0f} reference-equivalent TFLOP/s", flush=True)
