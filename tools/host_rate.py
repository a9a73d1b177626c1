"""Host-array-to-host-array rate of predict() and its phases (run on the GPU box).
usage: python tools/host_rate.py [edge] [dtype] [copy_threads]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from aind_exaspim_neuron_segmentation_amd import inference
from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D
from aind_exaspim_neuron_segmentation_amd.utils import synthetic

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cdt = sys.argv[2] if len(sys.argv) > 2 else "fp16"
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sd = synthetic.synth_state_dict(3, 1, seed=1)
m = UNet3D(output_channels=3, compute_dtype=cdt)
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
m.to("cuda").eval()
vol = np.random.default_rng(0).integers(0, 2000, (edge,) * 3, dtype=np.uint16)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = inference.predict(vol, m, verbose=False, return_device_tensor=True)
    torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
    del out
    tm = {}
    t0 = time.perf_counter()
    out = inference.predict_streaming(vol, m, verbose=False, copy_threads=threads, timings=tm)
    t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    out2 = inference.predict_streaming(vol, m, verbose=False, copy_threads=threads)
    t_host2 = time.perf_counter() - t0
    del out2
    t0 = time.perf_counter()
    out2 = inference.predict_streaming(vol, m, verbose=False, copy_threads=threads, out_dtype=np.float16)
    t_half = time.perf_counter() - t0
    print(f"{edge}^3 {cdt}: device-resident {t_dev:.3f} s ({edge**3/t_dev:.3e} vox/s) | host->host "
          f"{t_host2:.3f} s ({edge**3/t_host2:.3e} vox/s) | instrumented {t_host:.3f} s: "
          + ", ".join(f"{k} {v:.3f}" for k, v in tm.items())
          + f" | host->host float16 export {t_half:.3f} s", flush=True)
    del out, out2
