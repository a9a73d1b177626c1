import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D
from aind_exaspim_neuron_segmentation_amd.utils import synthetic
dev = torch.device("cuda:0")
for cdt in ("bf16", "fp32"):
    sd = synthetic.synth_state_dict(3, 1, seed=1)
    m = UNet3D(3, compute_dtype=cdt); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}); m.to(dev).eval()
    torch.manual_seed(0)
    x = torch.rand((2, 1, 96, 96, 96), device=dev)
    full = m.run(x, apply_sigmoid=True)
    out = torch.full_like(full, -7.0)
    part = m.run(x, apply_sigmoid=True, out=out, trim=8)
    inner = (Ellipsis,) + (slice(8, -8),) * 3
    d = (part[inner] != full[inner])
    idx = d.nonzero().cpu().numpy()
    print(cdt, "differing:", len(idx), "of", d.numel())
    if len(idx):
        for ax, name in enumerate("ncZYX"):
            vals, cnts = np.unique(idx[:, ax], return_counts=True)
            print("  axis", name, dict(zip((vals + (8 if ax >= 2 else 0)).tolist()[:40], cnts.tolist()[:40])))
        i = tuple(idx[0]); print("  first", i, float(part[inner][i]), float(full[inner][i]))
    part2 = m.run(x, apply_sigmoid=True, out=torch.full_like(full, -7.0), trim=8)
    print("  trimmed run repeatable:", bool(torch.equal(part2, part)))
