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
vol = synthetic.synth_volume((512, 512, 512), seed=5)
inference.predict(vol[:160, :160, :160], model, verbose=False)  # warm-up
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    out = inference.predict(vol, model, verbose=False)
    t1 = time.time()
    torch.cuda.synchronize(); t2 = time.time()
    dv = inference.predict(vol, model, verbose=False, return_device_tensor=True)
    torch.cuda.synchronize(); t3 = time.time()
    print(f"512^3 host->host {t1 - t0:.3f} s ({vol.size / (t1 - t0):.3e} vox/s), host->device result {t3 - t2:.3f} s, out {out.dtype} {out.shape}")
