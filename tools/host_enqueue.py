import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from aind_exaspim_neuron_segmentation_amd import inference
from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D
from aind_exaspim_neuron_segmentation_amd.utils import synthetic
dev = torch.device('cuda:0')
sd = synthetic.synth_state_dict(3, 1, seed=1)
model = UNet3D(output_channels=3, compute_dtype='bf16')
model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
model = model.to(dev).eval()
vol = inference.DeviceVolume.from_array(synthetic.synth_volume((512, 512, 512), seed=5), dev)
plan = inference.SlidingWindow(vol.shape, (96, 96, 96), (32, 32, 32), 8)
mn, mx = inference.volume_percentiles(vol, 1000, (1, 99.9))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    acc = inference.run_sliding_window(vol, model, plan, 3, 16, 1000, mn, mx)
    t1 = time.time()
    torch.cuda.synchronize(); t2 = time.time()
    print(f"enqueue {t1 - t0:.3f} s, total {t2 - t0:.3f} s for {len(plan.starts())} patches ({len(plan.starts()) // 16} batches)")
