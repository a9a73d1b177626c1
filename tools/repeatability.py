"""Four identical forward passes (full and trimmed, fp16 and fp32) must agree bit for bit: run on the MI355X box
after touching a kernel's stores or waits (it caught the gfx950 store-data hazard, profiles/r03_gfx950_hazards.txt)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D
from aind_exaspim_neuron_segmentation_amd.utils import synthetic
dev = torch.device("cuda:0")
for cdt in ("fp16", "fp32"):
    sd = synthetic.synth_state_dict(3, 1, seed=1)
    m = UNet3D(3, compute_dtype=cdt); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}); m.to(dev).eval()
    torch.manual_seed(0)
    x = torch.rand((4, 1, 96, 96, 96), device=dev)
    for trim in (0, 8):
        outs = [m.run(x, apply_sigmoid=True, out=torch.full((4, 3, 96, 96, 96), -7.0, device=dev), trim=trim).clone() for _ in range(4)]
        nd = [int((o != outs[0]).sum()) for o in outs[1:]]
        print(cdt, "trim", trim, "voxels differing from the first of 4 runs:", nd)
