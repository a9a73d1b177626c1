import sys, numpy as np, torch
sys.path.insert(0, ".")
from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D
from aind_exaspim_neuron_segmentation_amd.utils import synthetic
sd = synthetic.synth_state_dict(3, 1, seed=1)
for cdt in ("fp32", "fp16"):
    m = UNet3D(output_channels=3, compute_dtype=cdt)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    m.to("cuda").eval()
    for shape, trim in (((16,16,16),7), ((16,16,16),5), ((16,16,16),3), ((32,32,32),5), ((32,32,32),7), ((32,32,32),3), ((48,48,48),5)):
        x = torch.rand((2,1)+shape, device="cuda")
        full = m.run(x, apply_sigmoid=False)
        part = m.run(x, apply_sigmoid=False, trim=trim)
        inner = (Ellipsis,) + (slice(trim, -trim),)*3
        d = (part[inner]-full[inner]).abs()
        idx = torch.nonzero(d > 0)
        print(cdt, shape, trim, "max diff", float(d.max()), "n diff", int((d>0).sum()), "of", d.numel(), idx[:4].tolist())
