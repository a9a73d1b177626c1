"""Device-resident 1024^3-style steps with 1..6 batches in flight (run on the MI355X box):
    python tools/stream_sweep.py [--size 1024] [--dtype fp16] [--batch 16]"""
import argparse
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

p = argparse.ArgumentParser()
p.add_argument("--size", type=int, default=1024)
p.add_argument("--dtype", default="fp16")
p.add_argument("--batch", type=int, default=16)
a = p.parse_args()
dev = torch.device("cuda:0")
sd = synthetic.synth_state_dict(3, 1, seed=1)
model = UNet3D(output_channels=3, compute_dtype=a.dtype)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model.to(dev).eval()
shape = (a.size,) * 3
vol_t = torch.empty(shape, dtype=torch.int16, device=dev)
blk = _native.Block.make(shape, (0, 0, 0), shape)
_native.check(_native.lib().exaspim_synth_volume_u16(vol_t.data_ptr(), blk, 0, None), "synth")
volume = inference.DeviceVolume(vol_t, np.uint16, (0, 0, 0), shape)
plan = inference.SlidingWindow(shape, (96, 96, 96), (32, 32, 32), 8)
accum = torch.zeros((3,) + shape, dtype=torch.float32, device=dev)
for streams in (1, 2, 3, 4, 5, 6, 3, 1):
    for rep in range(2):
        accum.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        inference.run_sliding_window(volume, model, plan, 3, a.batch, 1000, 19.0, 1000.0, accum=accum, n_streams=streams)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"n_streams {streams}: {dt * 1e3:8.1f} ms per step  {a.size ** 3 / dt:.4g} voxels/s", flush=True)
