"""Times the imported reference predict() next to oracle.predict() on the same input, in the build
container only (the reference is a read-only mount at /root/reference and never travels):
    PYTHONDONTWRITEBYTECODE=1 python tools/oracle_vs_reference.py [edge] [batch]
The oracle is bench.py's cpu_baseline (kind "port"): BASELINE.md wants the two within +-10 %."""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
for _name in ["kimimaro", "waterz", "gcsfs", "s3fs", "tifffile", "zarr", "google", "google.cloud",
              "google.cloud.storage"]:
    sys.modules[_name] = types.ModuleType(_name)
_fr = types.ModuleType("fastremap")
for _n in ("mask_except", "renumber", "unique"):
    setattr(_fr, _n, None)
sys.modules["fastremap"] = _fr
sys.path.insert(0, "/root/reference/src")

import torch  # noqa: E402
from aind_exaspim_neuron_segmentation import inference as ref_inf  # noqa: E402
from aind_exaspim_neuron_segmentation.machine_learning.unet3d import UNet3D as RefUNet3D  # noqa: E402

from aind_exaspim_neuron_segmentation_amd.utils import synthetic  # noqa: E402
from oracle import reference_path as oracle  # noqa: E402

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 160
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sd = synthetic.synth_state_dict(3, 1, seed=1)
ref = RefUNet3D(output_channels=3)
ref.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
ref.eval()
vol = synthetic.synth_volume((edge,) * 3, seed=0)
orc = oracle.OracleModel(sd)
for rep in range(4):      # alternating order: whichever runs second in a pair tends to be a little slower
    runs = [("reference", lambda: ref_inf.predict(vol, ref, batch_size=batch, verbose=False)),
            ("oracle", lambda: oracle.predict(vol, orc, batch_size=batch))]
    if rep % 2:
        runs.reverse()
    took, out = {}, {}
    for name, fn in runs:
        t0 = time.perf_counter()
        out[name] = fn()
        took[name] = time.perf_counter() - t0
    print(f"rep {rep} ({runs[0][0]} first): reference {took['reference']:.2f} s, oracle {took['oracle']:.2f} s, "
          f"ratio {took['oracle'] / took['reference']:.3f}, "
          f"max|diff| {float(np.abs(out['reference'] - out['oracle']).max()):.1e}, threads {torch.get_num_threads()}")
