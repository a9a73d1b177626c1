"""
Benchmark of the hot path: affinity voxels/sec of the sliding-window 3D-UNet
prediction over a synthetic volume resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
         --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full pass of predict()'s device pipeline over the volume:
histogram -> percentiles -> for every batch of patches (gather+normalise ->
U-Net -> sigmoid -> trimmed overlap-add) -> divide by coverage. Input (uint16)
and output (float32 x3) stay in HBM; no host copies inside the timed region
(the PCIe-inclusive rate is reported separately in DESIGN.md).

Workload at N = 1: BASELINE.json configs[2] -- 1024^3 volume, 96^3 patches,
overlap 32, trim 8, batch 16, bf16 activations / fp32 accumulation. At N > 1
the volume grows with N (weak scaling, 1024^3 per GPU): the global patch grid
is partitioned by sub-volume over a (z, y) rank grid and only the 16-voxel
output overlap bands travel between neighbours (RCCL send/recv over xGMI).

Rank 0 prints ONE JSON line with the contract fields plus "roofline" (dominant
kernel, HIP-event timed on the launch stream) and "cpu_baseline" (the CPU
oracle timed on this host's cores on a bounded sample).
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from aind_exaspim_neuron_segmentation_amd import _native, inference, sharding  # noqa: E402
from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D  # noqa: E402
from aind_exaspim_neuron_segmentation_amd.utils import synthetic  # noqa: E402

FLOP_PER_PATCH_96 = 370_145_230_848  # 2 x MAC over the 19 convs
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}  # dense MFMA, MI355X_MICROARCH.md
# Launches of the dominant kernel symbol, conv3x3x3_zpipe<.., 6, 8, 16, 2, 4, HEAD = 0, POOL = false>
# (32-cout decoder layers without the fused head; inc.3 is the POOL = true instantiation):
# (bit in the timing mask, Cin, Cout, edge)
DOMINANT_CONVS = [(14, 64, 32, 48), (15, 64, 32, 96)]  # up3.3, up4.0
TRIM = 8  # predict()'s default (inference.py:38)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    p.add_argument("--size", type=int, default=1024, help="volume edge per GPU")
    p.add_argument("--batch", type=int, default=16)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--streams", type=int, default=1,
                   help="batches in flight on separate HIP streams (default 1: kernels never share "
                        "the device, so the per-kernel roofline timing means what it says)")
    p.add_argument("--cpu-sample", type=int, default=160, help="edge of the CPU sample volume")
    return p.parse_args()


def cpu_baseline(sample_edge, full_edge):
    """Times the CPU oracle (torch CPU fp32, all host cores) on a bounded sample."""
    from oracle import reference_path as oracle

    sd = synthetic.synth_state_dict(3, 1, seed=1)
    model = oracle.OracleModel(sd)
    vol = synthetic.synth_volume((sample_edge,) * 3, seed=0)
    n_patches = oracle.count_patches((1, 1) + vol.shape, (96,) * 3, (32,) * 3)
    t0 = time.perf_counter()
    oracle.predict(vol, model, batch_size=8)
    dt = time.perf_counter() - t0
    full_patches = oracle.count_patches((1, 1, full_edge, full_edge, full_edge), (96,) * 3, (32,) * 3)
    t_pre0 = time.perf_counter()
    oracle.normalize(np.minimum(synthetic.synth_volume((256,) * 3, seed=0), 1000))
    t_pre = (time.perf_counter() - t_pre0) * (full_edge / 256.0) ** 3
    est = full_patches * (dt / n_patches) + t_pre
    return {
        "value": float(full_edge) ** 3 / est,
        "unit": "voxels/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": (
            f"oracle predict() on {sample_edge}^3 ({n_patches} patches, batch 8) took {dt:.1f} s "
            f"= {dt / n_patches:.2f} s/patch; extrapolated to {full_edge}^3 = {full_patches} patches "
            f"+ {t_pre:.0f} s normalise (scaled from 256^3); os.cpu_count()={os.cpu_count()}"
        ),
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one rank per GPU; EXASPIM_DIST_BACKEND=gloo lets several ranks rehearse the
    # sharded path on a single GPU (device index wraps, transfers staged via host)
    backend = os.environ.get("EXASPIM_DIST_BACKEND", "nccl")
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    group = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    # model: random-init weights of the reference architecture (no checkpoints offline)
    sd = synthetic.synth_state_dict(3, 1, seed=1)
    model = UNet3D(output_channels=3, compute_dtype=args.dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(device).eval()

    # synthetic global volume, weak scaling over a (z, y) rank grid
    grid = sharding.rank_grid(world)
    gshape = (args.size * grid[0], args.size * grid[1], args.size)
    plan = inference.SlidingWindow(gshape, (96, 96, 96), (32, 32, 32), 8)
    shard = sharding.Shard(plan, grid, rank)
    vol_t = torch.empty(shard.input_dims, dtype=torch.int16, device=device)
    blk = _native.Block.make(shard.input_dims, shard.input_origin, gshape)
    _native.check(
        _native.lib().exaspim_synth_volume_u16(vol_t.data_ptr(), blk, 0, None), "synth"
    )
    volume = inference.DeviceVolume(vol_t, np.uint16, shard.input_origin, gshape)
    torch.cuda.synchronize()

    def step():
        return sharding.predict_shard(
            volume, model, plan, shard, n_channels=3, batch_size=args.batch,
            brightness_clip=1000, normalization_percentiles=(1, 99.9), group=group,
            n_streams=args.streams,
        )

    def barrier():
        if group is not None:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.device(device):
        model._ensure_engine(device)  # pack + upload weights before anything is timed
    for _ in range(args.warmup):
        out = step()
        del out
    lib = _native.lib()
    mask = 0
    for bit, _, _, _ in DOMINANT_CONVS:
        mask |= 1 << bit
    barrier()
    _native.check(lib.exaspim_unet_timing_begin(model._engine, mask), "timing_begin")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    import ctypes

    ms = (ctypes.c_double * 17)()
    cnt = (ctypes.c_int32 * 17)()
    _native.check(lib.exaspim_unet_timing_read(model._engine, ctypes.byref(ms), ctypes.byref(cnt)),
                  "timing_read")
    checksum = float(out.sum().item())
    if group is not None:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_vox = float(gshape[0]) * gshape[1] * gshape[2]
        value = total_vox * args.steps / elapsed
        # roofline of the dominant kernel symbol (the 32-cout MFMA convolution: up3.3 and
        # up4.0 are the same instantiation, so rocprofv3's per-kernel average covers
        # exactly these launches)
        launches = sum(cnt[b] for b, _, _, _ in DOMINANT_CONVS)
        k_ms = sum(ms[b] for b, _, _, _ in DOMINANT_CONVS)
        flops = 0.0
        for b, cin, cout, edge in DOMINANT_CONVS:
            # up4.0 (bit 15) only computes what up4.3 reads of the voxels predict() keeps:
            # trim - 1 voxels less on every face (exaspim_unet_forward_trimmed)
            need = edge - 2 * (TRIM - 1) if b == 15 else edge
            flops += cnt[b] * 2.0 * 27 * cin * cout * args.batch * need ** 3
        # the last batch of a step may be short; scale by the real patch count
        patches_per_step = len(shard.starts)
        full_batches = -(-patches_per_step // args.batch)
        flops *= patches_per_step / float(full_batches * args.batch)
        achieved = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.dtype]
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC
        # passes (FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md says;
        # profiles/summarize.py); null when no pass exists for this dtype.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"r01_pmc_hbm_{args.dtype}.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                kernels = json.load(f).get("kernels", {})
            tag = {"bf16": "BF16Tag", "fp16": "F16Tag", "fp32": "F32Tag"}[args.dtype]
            entry = kernels.get(f"conv3x3x3_zpipe<{tag}, 6, 8, 16, 2, 4, 0, false>")
            if entry:
                traffic = entry["hbm_bytes_per_launch"]
        result = {
            "metric": "affinity voxels/sec on 96^3 patches over a 1024^3 volume",
            "value": value,
            "unit": "voxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{gshape[0]}x{gshape[1]}x{gshape[2]} uint16 volume, 96^3 patches, "
                            f"overlap 32, trim 8, batch {args.batch}, "
                            f"{args.dtype} activations / fp32 accumulate (BASELINE.json configs[2] per GPU)",
                "patches_per_step": patches_per_step * world,
                "rank_grid_zy": list(grid),
                "sharding": "global patch grid partitioned by sub-volume; 16-voxel output bands "
                            "exchanged between neighbours" if world > 1 else "single device",
                "tflops_end_to_end": value * FLOP_PER_PATCH_96 / 64 ** 3 / 1e12,
                "output_checksum": checksum,
                "streams": args.streams,
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "conv3x3x3_zpipe<tile 6x8x16, 32 couts, no head, no pool> (launches: up3.3, up4.0)",
                "algorithmic_flop_per_launch": flops / launches if launches else None,
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "avg_launch_ms": k_ms / launches if launches else None,
                "launches": launches,
                "traffic": traffic,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.size)
        print(json.dumps(result))
    if group is not None:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
