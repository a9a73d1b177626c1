"""
Benchmark of the hot path: affinity voxels/sec of the sliding-window 3D-UNet
prediction over a synthetic volume.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it
starts N fresh child processes of this script (one rank per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set) before it
touches the GPU in any way, waits for them and exits non-zero if one fails.
Under `python -m torch.distributed.run ... bench.py --gpus N` the ranks exist
already and each process is one of them.

A "step" is one full pass of predict()'s device pipeline over the volume:
histogram -> percentiles -> for every batch of patches (gather+normalise ->
U-Net -> sigmoid -> trimmed overlap-add) -> divide by coverage. `value` keeps
input (uint16) and output (float32 x3) in HBM; the host-array-to-host-array
rate of the same call (upload, slab-wise download overlapped with compute) is
reported next to it as `host_to_host` at N = 1.

Workloads (BASELINE.json configs, --size scales the 1024 edge):
  N = 1  1024 x 1024 x 1024              configs[2], batch 16, 16-bit storage / fp32 accumulate
  N = 2  2048 x 1024 x 1024, grid 2 x 1  configs[3]
  N = 4  2048 x 2048 x 1024, grid 2 x 2
  N = 8  4096 x 2048 x 2048, grid 4 x 2  configs[4]
The global patch grid is partitioned by sub-volume over a (z, y) rank grid and
only the 16-voxel output overlap bands travel between neighbours (RCCL
send/recv over xGMI), plus a 512 KiB histogram all-reduce.

16-bit mode: the default --dtype is fp16 (IEEE half storage of activations and
weights, saturating stores, fp32 accumulation on v_mfma_f32_32x32x16_f16): it
meets north_star's 1e-3 against the reference (measured 2.7e-4 max, `parity`
block); bf16 storage runs at the same speed but sits at 2.3e-3 .. 2.9e-3.

Rank 0 prints ONE JSON line with the contract fields plus "roofline" (dominant
kernel, HIP-event timed on the launch stream), "parity" (this dtype against the
reference's own 160^3 output, tests/golden/g6), "pipelined" (the same steps with
three batches in flight on separate HIP streams -- what predict() does by default:
faster, but kernels overlap, so it is reported beside `value`, never as it),
"host_to_host", "configs" (at N = 1 the other single-GPU configurations of
BASELINE.json: configs[1] = 512^3, batch 8, fp32, and configs[2] in its literal
bf16 wording, each with value, dominant-kernel fraction and parity) and
"cpu_baseline" (the CPU oracle timed on this host's cores on a bounded sample).

N > 1: every rank synthesises only its DISJOINT sub-volume and fetches the halo its
last patches read from the +z / +y / +z+y neighbours inside the timed step
(--input-halo exchange, the default; "synth" synthesises the halo in place);
config.exchange_ms is split into histogram_ms / input_halo_ms / output_bands_ms.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_PATCH_96 = 370_145_230_848  # 2 x MAC over the 19 convs
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}  # dense MFMA, MI355X_MICROARCH.md
# Launches of the dominant kernel symbol, conv3x3x3_zpipe<.., 6, 8, 16, 2, 4, HEAD = 0, POOL = false>
# (32-cout decoder layers without the fused head; inc.3 is the POOL = true instantiation):
# (bit in the timing mask, Cin, Cout, edge)
DOMINANT_CONVS = [(14, 64, 32, 48), (15, 64, 32, 96)]  # up3.3, up4.0
TRIM = 8  # predict()'s default (inference.py:38)
TIMER_RING = 16384  # launches the engine's event ring holds (csrc/engine.hip)
PROFILE_ROUND = "r03"  # profiles/<round>_pmc_hbm_<dtype>.json of the code this file measures
# (z, y, x) extents in units of --size per world size: BASELINE.json configs[2], [3], -, [4]
GLOBAL_SHAPES = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (4, 2, 2)}
CONFIG_NAMES = {1: "configs[2]", 2: "configs[3]", 8: "configs[4]"}


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp32"])
    p.add_argument("--size", type=int, default=1024, help="edge of the per-GPU cube (1024 = BASELINE)")
    p.add_argument("--batch", type=int, default=16)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-host-to-host", action="store_true")
    p.add_argument("--no-parity", action="store_true")
    p.add_argument("--streams", type=int, default=1,
                   help="batches in flight on separate HIP streams (default 1: kernels never share "
                        "the device, so the per-kernel roofline timing means what it says)")
    p.add_argument("--pipelined-streams", type=int, default=3,
                   help="batches in flight of the extra 'pipelined' measurement (0 = skip it)")
    p.add_argument("--cpu-sample", type=int, default=224, help="edge of the CPU sample volume")
    p.add_argument("--cpu-batch", type=int, default=16, help="batch size of the CPU sample run")
    p.add_argument("--no-configs", action="store_true", help="skip the extra single-GPU config legs")
    p.add_argument("--input-halo", default=None, choices=["exchange", "synth"],
                   help="N > 1: fetch the input halo from the neighbours inside the timed step "
                        "(exchange, default) or synthesise every rank's halo in place (synth)")
    return p.parse_args()


def zcol_main(ext, tile):
    """Extent the z-column kernel covers with whole tiles (csrc/conv3d.hip:
    conv_zcol_main_extent): a remainder of 1..4 voxels goes to thin-tile launches."""
    rem = ext % tile
    return ext - rem if ext > tile and 1 <= rem <= 4 else ext


def launch_ranks(args):
    """Parent of an N-rank run: spawns the ranks and never initialises the GPU. The children are
    polled; the first one that exits non-zero takes the others down with it (they would otherwise
    sit in a collective waiting for the dead rank) and the parent exits non-zero."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    bad = []
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad or all(c == 0 for c in codes):
                break
            time.sleep(0.2)
    finally:
        live = [p for p in procs if p.poll() is None]     # exactly the children started above
        for p in live:
            p.terminate()
        for p in live:
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    if bad:
        raise SystemExit(f"bench.py: ranks failed (rank, exit code): {bad}; the other ranks were stopped")


def cpu_baseline(sample_edge, full_edge, batch):
    """Times the CPU oracle (torch CPU fp32, all host cores) on a bounded sample."""
    import numpy as np
    import torch

    from aind_exaspim_neuron_segmentation_amd.utils import synthetic
    from oracle import reference_path as oracle

    sd = synthetic.synth_state_dict(3, 1, seed=1)
    model = oracle.OracleModel(sd)
    vol = synthetic.synth_volume((sample_edge,) * 3, seed=0)
    n_patches = oracle.count_patches((1, 1) + vol.shape, (96,) * 3, (32,) * 3)
    t0 = time.perf_counter()
    oracle.predict(vol, model, batch_size=batch)
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
            f"oracle predict() on {sample_edge}^3 ({n_patches} patches, batch {batch}) took {dt:.1f} s "
            f"= {dt / n_patches:.2f} s/patch (measured); extrapolated to {full_edge}^3 = {full_patches} patches "
            f"+ {t_pre:.0f} s normalise (scaled from a measured 256^3); os.cpu_count()={os.cpu_count()}"
        ),
    }


def parity_block(model, dtype):
    """The benchmarked model (same weights, same compute dtype) on the reference's
    default-config 160^3 case against the reference's own output (golden g6)."""
    import numpy as np

    from aind_exaspim_neuron_segmentation_amd import inference
    from aind_exaspim_neuron_segmentation_amd.utils import synthetic

    path = os.path.join(ROOT, "tests", "golden", "g6_default_160.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path, allow_pickle=False)
    vol = synthetic.synth_volume((160, 160, 160), seed=0)
    got = inference.predict(vol, model, batch_size=8, verbose=False)
    err = np.abs(got[:, ::5, ::5, ::5] - g["pred_sub"])
    return {
        "against": "reference predict() on 160^3, defaults (tests/golden/g6_default_160.npz)",
        "dtype": dtype,
        "max": float(err.max()),
        "mean": float(err.mean()),
        "p99_9": float(np.quantile(err, 0.999)),
        "tolerance": 1e-3,
        "weights": "seeded random (no checkpoint ships with the reference)",
    }


def host_to_host(model, args, edge):
    """predict() from a host numpy array to a host numpy array on the N = 1 workload, with
    predict()'s own defaults (three batches in flight) and, beside it, on a single stream."""
    import numpy as np
    import torch

    from aind_exaspim_neuron_segmentation_amd import inference

    dev = next(model.parameters()).device
    vol = synth_block((0, 0, 0), (edge,) * 3, (edge,) * 3, dev).cpu().numpy().view(np.uint16)
    torch.cuda.empty_cache()
    times = []
    out = None
    for _ in range(3):          # the first call also page-locks the staging buffers
        del out                 # (unmapping the previous 12 B/voxel result is not part of a call)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = inference.predict(vol, model, batch_size=args.batch, verbose=False)
        times.append(time.perf_counter() - t0)
    dt = min(times[1:])
    checksum = float(out[:, ::7, ::7, ::7].sum(dtype=np.float64))
    phases = {}
    inference.predict_streaming(vol, model, batch_size=args.batch, verbose=False, timings=phases)
    single = None
    if args.pipelined_streams > 1:
        # the same call on one stream (bit-identical result)
        stimes = []
        for _ in range(2):
            del out
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = inference.predict(vol, model, batch_size=args.batch, verbose=False, n_streams=1)
            stimes.append(time.perf_counter() - t0)
        single = {"streams": 1, "value": float(edge) ** 3 / min(stimes), "ms_per_step": min(stimes) * 1e3,
                  "output_checksum": float(out[:, ::7, ::7, ::7].sum(dtype=np.float64))}
    return {
        "phases_s": {k: round(v, 4) for k, v in phases.items()},
        "value": float(edge) ** 3 / dt,
        "unit": "voxels/s",
        "ms_per_step": dt * 1e3,
        "streams": inference.DEFAULT_STREAMS,
        "single_stream": single,
        "first_call_ms": times[0] * 1e3,
        "what": "inference.predict(numpy uint16 -> numpy float32 (3, D, H, W)) with its defaults: chunked "
                "upload + histogram, finished slabs downloaded to pinned memory on a copy stream and "
                "moved into the pageable result by 4 host threads while later layers compute",
        "output_checksum": checksum,
    }


def make_model(dtype, device):
    """Random-init weights of the reference architecture (no checkpoints offline)."""
    import torch

    from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D
    from aind_exaspim_neuron_segmentation_amd.utils import synthetic

    sd = synthetic.synth_state_dict(3, 1, seed=1)
    model = UNet3D(output_channels=3, compute_dtype=dtype)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.to(device).eval()
    with torch.cuda.device(device):
        model._ensure_engine(device)  # pack + upload weights before anything is timed
    return model


def synth_block(origin, dims, gshape, device):
    """uint16 block [origin, origin + dims) of the synthetic global volume, generated on the device."""
    import torch

    from aind_exaspim_neuron_segmentation_amd import _native

    t = torch.empty(tuple(dims), dtype=torch.int16, device=device)
    blk = _native.Block.make(tuple(dims), tuple(origin), tuple(gshape))
    _native.check(_native.lib().exaspim_synth_volume_u16(t.data_ptr(), blk, 0, None), "synth")
    return t


def dominant_roofline(ms, cnt, dtype, batch, patches_per_step):
    """Roofline block of the dominant kernel symbol from the engine's event timers (the 32-cout
    MFMA convolution: up3.3 and up4.0 are the same instantiation, so rocprofv3's per-kernel
    average covers exactly these launches)."""
    launches = sum(cnt[b] for b, _, _, _ in DOMINANT_CONVS)
    k_ms = sum(ms[b] for b, _, _, _ in DOMINANT_CONVS)
    flops = 0.0
    issued = 0.0    # what the launches' whole 6 x 8 x 16 tiles compute (masked planes included)
    for b, cin, cout, edge in DOMINANT_CONVS:
        # up4.0 (bit 15) only computes what up4.3 reads of the voxels predict() keeps:
        # trim - 1 voxels less on every face (exaspim_unet_forward_trimmed) = 82^3; of
        # that the timed z-column launch covers the whole 8 x 16 (y, x) tiles, 82 x 80 x 80
        # (the two 2-voxel-thick remainders run as separate, untimed thin-tile launches)
        need = edge - 2 * (TRIM - 1) if b == 15 else edge
        vox = need * zcol_main(need, 8) * zcol_main(need, 16) if b == 15 else need ** 3
        flops += cnt[b] * 2.0 * 27 * cin * cout * batch * vox
        tiles = (-(-need // 6)) * (-(-(zcol_main(need, 8) if b == 15 else need) // 8)) * \
                (-(-(zcol_main(need, 16) if b == 15 else need) // 16))
        issued += cnt[b] * 2.0 * 27 * cin * cout * batch * tiles * (6 * 8 * 16)
    # the last batch of a step may be short; scale by the real patch count
    full_batches = -(-patches_per_step // batch)
    flops *= patches_per_step / float(full_batches * batch)
    issued *= patches_per_step / float(full_batches * batch)
    achieved = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    peak = PEAK_TFLOPS[dtype]
    launches_per_step = len(DOMINANT_CONVS) * full_batches
    tag = {"bf16": "BF16Tag", "fp16": "F16Tag", "fp32": "F32Tag"}[dtype]
    # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes of THIS
    # round's code (FETCH_SIZE / WRITE_SIZE, corrected as MI355X_MICROARCH.md says;
    # profiles/summarize.py); null when no such pass exists for this dtype.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_hbm_{dtype}.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            entry = json.load(f).get("kernels", {}).get(f"conv3x3x3_zpipe<{tag}, 6, 8, 16, 2, 4, 0, false>")
        if entry:
            traffic = entry["hbm_bytes_per_launch"]
    return {
        "bound": "mfma",
        "kernel": f"conv3x3x3_zpipe<{tag}, tile 6x8x16, 32 couts, no head, no pool> (launches: up3.3, up4.0)",
        "algorithmic_flop_per_launch": flops / launches if launches else None,
        "issued_flop_per_launch": issued / launches if launches else None,
        "achieved": achieved,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": achieved / peak,
        "avg_launch_ms": k_ms / launches if launches else None,
        "timed_launches": launches,
        "steps_covered": launches / float(launches_per_step) if launches_per_step else None,
        "timer_ring": TIMER_RING,
        "traffic": traffic,
    }


def timed_steps(model, step, steps, warmup, barrier):
    """W untimed + K timed steps bracketed by barrier(); returns (seconds, event ms, counts, last result)."""
    import ctypes

    from aind_exaspim_neuron_segmentation_amd import _native

    for _ in range(warmup):
        out = step()
        del out
    lib = _native.lib()
    mask = 0
    for bit, _, _, _ in DOMINANT_CONVS:
        mask |= 1 << bit
    barrier()
    _native.check(lib.exaspim_unet_timing_begin(model._engine, mask), "timing_begin")
    out = None
    t0 = time.perf_counter()
    for _ in range(steps):
        del out             # the consumer is done with a result before it asks for the next one
        out = step()        # (else every second step finds no cached 12.9 GB block: a hipMalloc)
    barrier()
    elapsed = time.perf_counter() - t0
    ms = (ctypes.c_double * 17)()
    cnt = (ctypes.c_int32 * 17)()
    _native.check(lib.exaspim_unet_timing_read(model._engine, ctypes.byref(ms), ctypes.byref(cnt)),
                  "timing_read")
    return elapsed, ms, cnt, out


def config_leg(name, dtype, edge, batch, device, steps=2, warmup=1):
    """One of BASELINE.json's other single-GPU configurations: device-resident steps on one
    stream, the dominant kernel's roofline fraction and the parity of this dtype."""
    import numpy as np
    import torch

    from aind_exaspim_neuron_segmentation_amd import inference, sharding

    model = make_model(dtype, device)
    gshape = (edge,) * 3
    plan = inference.SlidingWindow(gshape, (96, 96, 96), (32, 32, 32), TRIM)
    shard = sharding.Shard(plan, (1, 1), 0)
    volume = inference.DeviceVolume(synth_block((0, 0, 0), gshape, gshape, device), np.uint16, (0, 0, 0), gshape)

    def step():
        return sharding.predict_shard(volume, model, plan, shard, n_channels=3, batch_size=batch,
                                      brightness_clip=1000, normalization_percentiles=(1, 99.9), n_streams=1)

    elapsed, ms, cnt, out = timed_steps(model, step, steps, warmup, torch.cuda.synchronize)
    checksum = float(out.sum().item())
    del out, volume
    torch.cuda.empty_cache()
    value = float(edge) ** 3 * steps / elapsed
    leg = {
        "config": name,
        "workload": f"{edge}x{edge}x{edge} uint16 volume, 96^3 patches, overlap 32, trim 8, batch {batch}, {dtype}",
        "dtype": dtype,
        "value": value,
        "unit": "voxels/s",
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "reference_equivalent_tflops": value * FLOP_PER_PATCH_96 / 64 ** 3 / 1e12,
        "output_checksum": checksum,
        "roofline": dominant_roofline(ms, cnt, dtype, batch, len(shard.starts)),
        "parity": parity_block(model, dtype),
    }
    del model
    torch.cuda.empty_cache()
    return leg


def main():
    args = parse_args()
    if args.gpus not in GLOBAL_SHAPES:
        raise SystemExit(f"--gpus {args.gpus}: supported world sizes are {sorted(GLOBAL_SHAPES)}")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)      # no torch.cuda call has happened in this process
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import datetime

    import numpy as np
    import torch

    from aind_exaspim_neuron_segmentation_amd import inference, sharding

    # one rank per GPU; EXASPIM_DIST_BACKEND=gloo lets several ranks rehearse the
    # sharded path on a single GPU (device index wraps, transfers staged via host)
    backend = os.environ.get("EXASPIM_DIST_BACKEND", "nccl")
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    group = None
    if world > 1:
        import torch.distributed as dist

        # a rank that dies leaves its peers in a collective: bounded by this timeout, and the
        # launcher (launch_ranks) stops them as soon as it sees the dead rank
        timeout = datetime.timedelta(minutes=5)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=timeout)
        else:
            dist.init_process_group(backend, timeout=timeout)
        group = dist.group.WORLD
    if os.environ.get("EXASPIM_BENCH_FAIL_RANK") == str(rank):     # test hook: one rank dies alone
        raise SystemExit(f"bench.py: injected failure of rank {rank}")

    model = make_model(args.dtype, device)

    # synthetic global volume over a (z, y) rank grid
    grid = sharding.rank_grid(world)
    gshape = tuple(args.size * m for m in GLOBAL_SHAPES[world])
    plan = inference.SlidingWindow(gshape, (96, 96, 96), (32, 32, 32), TRIM)
    shard = sharding.Shard(plan, grid, rank)
    halo_mode = args.input_halo or "exchange"
    exchange_halo = world > 1 and halo_mode == "exchange"
    exchange = {"seconds": 0.0}
    if exchange_halo:
        # every rank holds its disjoint sub-volume only; the halo arrives inside the step
        core_t = synth_block(shard.core_origin, shard.core_dims, gshape, device)
        volume = None
    else:
        core_t = None
        volume = inference.DeviceVolume(synth_block(shard.input_origin, shard.input_dims, gshape, device),
                                        np.uint16, shard.input_origin, gshape)
    torch.cuda.synchronize()

    def step(n_streams=args.streams, timings=exchange):
        vol = volume
        if exchange_halo:
            block = sharding._timed(timings, "input_halo_s", device,
                                    lambda: sharding.exchange_input_halo(core_t, shard, group))
            vol = inference.DeviceVolume(block, np.uint16, shard.input_origin, gshape)
        return sharding.predict_shard(
            vol, model, plan, shard, n_channels=3, batch_size=args.batch,
            brightness_clip=1000, normalization_percentiles=(1, 99.9), group=group,
            n_streams=n_streams, timings=timings, core=core_t,
        )

    def barrier():
        if group is not None:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    def timed_step():
        return step()

    for _ in range(args.warmup):
        out = step()
        del out
    exchange.clear()
    exchange["seconds"] = 0.0
    elapsed, ms, cnt, out = timed_steps(model, timed_step, args.steps, 0, barrier)
    checksum = float(out.sum().item())
    del out
    # The same K steps with several batches in flight (predict(..., n_streams=k), predict()'s
    # default): kernels of different batches then share the device, one kernel's ramp-down overlaps
    # another batch's work, and a kernel's event-timed duration no longer measures that kernel --
    # which is why `value` and `roofline` above are single-stream. Reported next to them, never as
    # `value`.
    pipelined = None
    if world == 1 and args.pipelined_streams > 1 and args.streams == 1:
        def step_pipelined():
            return step(n_streams=args.pipelined_streams, timings={"seconds": 0.0})
        out = step_pipelined()      # the worker streams' workspaces are allocated here
        del out
        barrier()
        psteps = min(args.steps, 3)     # an extra: bounded whatever K the caller asked for
        out = None
        tp0 = time.perf_counter()
        for _ in range(psteps):
            del out
            out = step_pipelined()
        barrier()
        tp = time.perf_counter() - tp0
        pipelined = {
            "streams": args.pipelined_streams,
            "steps": psteps,
            "value": float(gshape[0]) * gshape[1] * gshape[2] * psteps / tp,
            "unit": "voxels/s",
            "ms_per_step": tp / psteps * 1e3,
            "output_checksum": float(out.sum().item()),
            "what": "same steps with predict()'s default n_streams: bit-identical result (stitching stays in "
                    "batch order on the caller's stream); per-kernel durations are not meaningful here",
        }
        del out
    per_step = 1e3 / max(args.steps, 1)
    phases = [exchange.get("seconds", 0.0) * per_step, exchange.get("histogram_s", 0.0) * per_step,
              exchange.get("input_halo_s", 0.0) * per_step, exchange.get("output_bands_s", 0.0) * per_step]
    if group is not None:
        import torch.distributed as dist

        t = torch.tensor([elapsed] + phases, dtype=torch.float64,
                         device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, phases = float(t[0].item()), [float(v) for v in t[1:].tolist()]
    exchange_ms, histogram_ms, input_halo_ms, output_bands_ms = phases

    if rank == 0:
        total_vox = float(gshape[0]) * gshape[1] * gshape[2]
        value = total_vox * args.steps / elapsed
        patches_per_step = len(shard.starts)
        # FLOPs of the work actually launched: the trimmed forward skips the margin of
        # up4.0 (82^3 of 96^3) and up4.3 (80^3 of 96^3)
        skipped = 2.0 * 27 * (64 * 32 * (96 ** 3 - 82 ** 3) + 32 * 32 * (96 ** 3 - 80 ** 3))
        storage = {"fp16": "fp16 (IEEE half) storage", "bf16": "bf16 storage", "fp32": "fp32"}[args.dtype]
        name = CONFIG_NAMES.get(world)
        shape_txt = "x".join(str(v) for v in gshape)
        if world > 1:
            sharding_txt = (
                "global patch grid partitioned by sub-volume over a (z, y) rank grid; per step a 512 KiB "
                "histogram all-reduce, the 16-voxel output bands to the +z / +y neighbours and "
                + ("the input halo FETCHED from the +z / +y / +z+y neighbours inside the timed step "
                   "(exchange_input_halo; every rank synthesises only its disjoint sub-volume)"
                   if exchange_halo else
                   "every rank's input block (sub-volume + halo) synthesised in place (--input-halo synth: "
                   "exchange_input_halo is not in the timed path)")
                + "; the halo is 32 voxels = the patch overlap, which is all a rank's last patches read beyond "
                  "its own sub-volume (BASELINE configs[3] words it as halo=48 = overlap + 2 x trim: the 16 "
                  "extra voxels would only feed outputs that the trim discards)")
        else:
            sharding_txt = "single device"
        result = {
            "metric": f"affinity voxels/sec on 96^3 patches over a {shape_txt} volume, {world} MI355X",
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
                "workload": f"{shape_txt} uint16 volume, 96^3 patches, overlap 32, trim 8, "
                            f"batch {args.batch}, {storage} of activations and weights / fp32 accumulate"
                            + (f" (BASELINE.json {name}" + (f", edge scaled to {args.size}" if args.size != 1024 else "") + ")"
                               if name else " (between BASELINE.json configs[3] and configs[4])"),
                "patches_per_step": patches_per_step * world,
                "rank_grid_zy": list(grid),
                "sharding": sharding_txt,
                "input_halo": ("exchange" if exchange_halo else "synth") if world > 1 else None,
                "exchange_ms": exchange_ms,
                "histogram_ms": histogram_ms,
                "input_halo_ms": input_halo_ms,
                "output_bands_ms": output_bands_ms,
                "reference_equivalent_tflops": value * FLOP_PER_PATCH_96 / 64 ** 3 / 1e12,
                "launched_tflops": value * (FLOP_PER_PATCH_96 - skipped) / 64 ** 3 / 1e12,
                "output_checksum": checksum,
                "streams": args.streams,
            },
            "roofline": dominant_roofline(ms, cnt, args.dtype, args.batch, patches_per_step),
        }
        if pipelined is not None:
            result["pipelined"] = pipelined
        if world == 1:
            if not args.no_parity:
                result["parity"] = parity_block(model, args.dtype)
            if not args.no_host_to_host:
                del volume
                torch.cuda.empty_cache()
                result["host_to_host"] = host_to_host(model, args, args.size)
            if not args.no_configs:
                # the other single-GPU configurations of BASELINE.json, so that one driver-run line
                # covers them: configs[1] in the parity-graded precision, configs[2] as literally worded
                volume = None
                del model
                torch.cuda.empty_cache()
                edge1 = max(96, args.size // 2)
                result["configs"] = {
                    "configs[1]": config_leg(
                        "512x512x512, batch 8, fp32, 1 GPU" + (f" (edge scaled to {edge1})" if edge1 != 512 else ""),
                        "fp32", edge1, 8, device),
                    "configs[2]_bf16": config_leg(
                        "1024x1024x1024, bf16 activations / fp32 accumulate, batch 16, 1 GPU (the literal wording; "
                        "the headline line runs the same workload with IEEE-half storage, the 16-bit mode inside 1e-3)"
                        + (f" (edge scaled to {args.size})" if args.size != 1024 else ""),
                        "bf16", args.size, 16, device),
                }
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.size, args.cpu_batch)
        print(json.dumps(result), flush=True)
    if group is not None:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
