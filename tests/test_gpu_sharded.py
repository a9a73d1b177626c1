"""
Sharded predict on the GPU box (-m gpu): 2 and 4 ranks share the one MI355X and
talk over gloo (device tensors staged through the host), which exercises every
line of sharding.predict_shard except the RCCL transport itself. The stitched
result must equal the single-process predict().
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aind_exaspim_neuron_segmentation_amd.utils import synthetic

pytestmark = pytest.mark.gpu

GSHAPE = (104, 88, 56)
# overlap 16, trim 4: stride 16, trimmed outputs of 24 -> 8-voxel partial-sum bands
# really cross the rank faces (with overlap 8 the trimmed outputs would tile exactly)
KW = dict(patch_shape=(32, 32, 32), overlap=(16, 16, 16), trim=4)


def _model(dev):
    from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D

    sd = synthetic.synth_state_dict(3, 1, seed=1)
    model = UNet3D(output_channels=3)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    return model.to(dev).eval()


def _worker(rank, world, port, failures):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from aind_exaspim_neuron_segmentation_amd import inference, sharding

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        model = _model(dev)
        gvol = synthetic.synth_volume(GSHAPE, seed=3)
        plan = inference.SlidingWindow(GSHAPE, KW["patch_shape"], KW["overlap"], KW["trim"])
        shard = sharding.Shard(plan, sharding.rank_grid(world), rank)
        # each rank holds only its disjoint sub-volume; the halo comes from the neighbours
        core_sl = tuple(slice(o, o + d) for o, d in zip(shard.core_origin, shard.core_dims))
        core = torch.from_numpy(np.ascontiguousarray(gvol[core_sl]).view(np.int16)).to(dev)
        block = sharding.exchange_input_halo(core, shard, dist.group.WORLD)
        volume = inference.DeviceVolume(block, np.uint16, shard.input_origin, GSHAPE)
        accum = sharding.predict_shard(volume, model, plan, shard, n_channels=3, batch_size=5,
                                       group=dist.group.WORLD)
        own = sharding.owned_result(accum, shard).cpu()
        parts = [None] * world
        dist.gather_object((shard.own_lo, shard.own_hi, own.numpy()), parts if rank == 0 else None, dst=0)
        if rank == 0:
            full = np.full((3,) + GSHAPE, np.nan, np.float32)
            for lo, hi, arr in parts:
                full[(slice(None),) + tuple(slice(a, b) for a, b in zip(lo, hi))] = arr
            assert not np.isnan(full).any()
            want = inference.predict(gvol, model, batch_size=5, verbose=False, **KW)
            err = np.abs(full - want).max()
            print(f"sharded x{world} vs single-process predict: max|diff| = {err:.3e}")
            assert err < 2e-6
            np.testing.assert_array_equal(full == 0, want == 0)
    except Exception as exc:
        failures.put(f"rank {rank}: {type(exc).__name__}: {exc}")
        raise
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_predict_matches_single_process(world):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    ctx = mp.get_context("spawn")
    failures = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, failures)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    msgs = []
    while not failures.empty():
        msgs.append(failures.get())
    assert not msgs, msgs
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (the driver's command form) must start the N ranks itself:
    the parent spawns the children before it touches the GPU and relays a non-zero exit.
    Rehearsed here with 2 ranks sharing the one GPU over gloo on a small volume; rank 0 prints
    the JSON line with the contract fields, a config-faithful workload string and exchange_ms."""
    import json
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EXASPIM_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(
        [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size", "128", "--steps", "1",
         "--warmup", "0", "--no-cpu-baseline"],
        env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["unit"] == "voxels/s" and d["value"] > 0
    assert "256x128x128" in d["metric"] and "256x128x128" in d["config"]["workload"]
    assert d["config"]["rank_grid_zy"] == [2, 1] and d["config"]["exchange_ms"] >= 0
    assert d["roofline"]["timed_launches"] > 0 and "cpu_baseline" not in d
    # north_star's exchange is inside the timed step: every rank holds its disjoint sub-volume and
    # fetches the halo from its neighbours; the three exchange phases are reported separately
    cfg = d["config"]
    assert cfg["input_halo"] == "exchange" and "exchange_input_halo" in cfg["sharding"]
    assert cfg["input_halo_ms"] > 0 and cfg["histogram_ms"] > 0 and cfg["output_bands_ms"] > 0
    assert abs(cfg["exchange_ms"] - (cfg["input_halo_ms"] + cfg["histogram_ms"] + cfg["output_bands_ms"])) \
        <= 0.5 * cfg["exchange_ms"] + 1.0      # (each figure is a max over ranks)
    synth = subprocess.run(
        [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size", "128", "--steps", "1",
         "--warmup", "0", "--no-cpu-baseline", "--input-halo", "synth"],
        env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert synth.returncode == 0, synth.stderr[-2000:]
    d2 = json.loads([ln for ln in synth.stdout.splitlines() if ln.startswith("{")][0])
    assert d2["config"]["input_halo"] == "synth" and d2["config"]["input_halo_ms"] == 0
    # the same volume either way: the result does not depend on where the halo came from
    assert d2["config"]["output_checksum"] == d["config"]["output_checksum"]
    # a failing rank makes the launcher exit non-zero
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size", "16"],   # no patch fits: Shard raises in every rank
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert bad.returncode != 0


def test_bench_on_a_2x2_rank_grid():
    """Four ranks (the 2 x 2 grid: z and y neighbours and the diagonal one) share the one GPU over gloo:
    the input halo arrives from three neighbours, bands travel along both axes and the corner is
    forwarded -- the same line fields, and the same output checksum as with the halo synthesised in place."""
    import json
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EXASPIM_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    sums = []
    for halo in ("exchange", "synth"):
        out = subprocess.run(
            [sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--size", "128", "--steps", "1",
             "--warmup", "0", "--no-cpu-baseline", "--input-halo", halo],
            env=env, capture_output=True, text=True, timeout=900, cwd=root)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        assert d["n_gpus"] == 4 and d["config"]["rank_grid_zy"] == [2, 2] and "256x256x128" in d["metric"]
        assert d["config"]["input_halo"] == halo and (d["config"]["input_halo_ms"] > 0) == (halo == "exchange")
        sums.append(d["config"]["output_checksum"])
    assert sums[0] == sums[1] and sums[0] > 0


def test_bench_stops_the_other_ranks_when_one_rank_dies():
    """Only rank 1 fails (test hook EXASPIM_BENCH_FAIL_RANK): rank 0 would sit in its first collective
    waiting for it. The launcher polls its children, stops the survivor and exits non-zero -- long
    before the process group's own timeout."""
    import subprocess
    import sys
    import time

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EXASPIM_DIST_BACKEND="gloo", EXASPIM_BENCH_FAIL_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    out = subprocess.run(
        [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--size", "128", "--steps", "1",
         "--warmup", "0", "--no-cpu-baseline"],
        env=env, capture_output=True, text=True, timeout=600, cwd=root)
    took = time.perf_counter() - t0
    assert out.returncode != 0 and "ranks failed" in out.stderr and "(1," in out.stderr, out.stderr[-1500:]
    assert took < 240, took
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_rccl_transport_rehearsal_on_one_rank():
    """The RCCL ("nccl") transport cannot run between ranks on a one-GPU box (RCCL refuses two
    ranks on one device), so what can be rehearsed is rehearsed at world size 1 in a child
    process: group creation the way bench.py does it, the histogram all-reduce on a device int64
    tensor, the barrier, and sharding._p2p's send/recv pair (to itself, inside one RCCL group
    call) with the tensors the two exchanges move: float32 partial sums and 16-bit voxels,
    which RCCL has no type for and which therefore travel as bytes."""
    import subprocess
    import sys

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, %r)
from aind_exaspim_neuron_segmentation_amd import sharding
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
g = dist.group.WORLD
h = torch.arange(65536, dtype=torch.int64, device=dev)
sharding.all_reduce_sum(h, g)
assert torch.equal(h.cpu(), torch.arange(65536))
dist.barrier()
for dtype in (torch.float32, torch.int16):
    a = (torch.arange(3 * 5 * 7, device=dev) - 50).to(dtype).reshape(3, 5, 7)
    b = torch.zeros_like(a)
    sharding._p2p([("send", a, 0), ("recv", b, 0)], g)
    torch.cuda.synchronize()
    assert torch.equal(a, b), dtype
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.tolist() == [1.5, 2.5]
dist.destroy_process_group()
print("rccl world-1 rehearsal ok")
""" % root
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    print(out.stdout[-600:], out.stderr[-1500:])
    assert out.returncode == 0 and "rehearsal ok" in out.stdout


def test_rank_block_of_the_8_gpu_workload_indexes_beyond_2_31():
    """One rank's share of BASELINE configs[4] (4096 x 2048 x 2048 over a 4 x 2 rank grid), run
    here without its neighbours: the input block of an interior rank is 1056 x 1056 x 2048 =
    2.28e9 voxels and its accumulators 6.9e9 floats, i.e. every voxel index of the gather,
    stitch and finalise kernels passes 2^31 -- sizes no other test reaches. The partial sums
    at the far corner of the block (largest indices) must equal, bit for bit, those of the
    same twelve patches run on a small block cut out of the same global volume."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from aind_exaspim_neuron_segmentation_amd import _native, inference, sharding
    from aind_exaspim_neuron_segmentation_amd.machine_learning.unet3d import UNet3D

    dev = torch.device("cuda:0")
    if torch.cuda.mem_get_info(dev)[0] < 60 * 2**30:
        pytest.skip("needs 60 GB of free device memory")
    sd = synthetic.synth_state_dict(3, 1, seed=1)
    model = UNet3D(output_channels=3, compute_dtype="fp16")
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    model.to(dev).eval()
    gshape = (4096, 2048, 2048)
    plan = inference.SlidingWindow(gshape, (96, 96, 96), (32, 32, 32), 8)
    shard = sharding.Shard(plan, sharding.rank_grid(8), 2)          # interior along z, first along y
    assert int(np.prod(shard.input_dims, dtype=np.int64)) > 2**31 and shard.input_origin[0] > 0
    assert len(shard.starts) == 8192

    def block_volume(origin, dims):
        t = torch.empty(dims, dtype=torch.int16, device=dev)
        blk = _native.Block.make(dims, origin, gshape)
        _native.check(_native.lib().exaspim_synth_volume_u16(t.data_ptr(), blk, 0, None), "synth")
        return inference.DeviceVolume(t, np.uint16, origin, gshape), blk

    mn, mx = 19.0, 1000.0
    volume, blk = block_volume(shard.input_origin, shard.input_dims)
    big = inference.run_sliding_window(volume, model, plan, 3, 16, 1000, mn, mx,
                                       starts=shard.starts, accum_block=blk)
    torch.cuda.synchronize()
    # the last two z and y starts and the last three x starts of the rank
    zs = sorted({s[0] for s in shard.starts})[-2:]
    ys = sorted({s[1] for s in shard.starts})[-2:]
    xs = sorted({s[2] for s in shard.starts})[-3:]
    sub = [(z, y, x) for z in zs for y in ys for x in xs]
    lo = (zs[0], ys[0], xs[0])
    hi = tuple(min(s[-1] + 96, g) for s, g in zip((zs, ys, xs), gshape))
    dims = tuple(h - l for l, h in zip(lo, hi))
    svol, sblk = block_volume(lo, dims)
    small = inference.run_sliding_window(svol, model, plan, 3, 16, 1000, mn, mx, starts=sub,
                                         accum_block=sblk)
    # voxels only these twelve patches write: 24 past the first start of the subset on every
    # axis (the previous patch's trimmed output ends at start - 64 + 88)
    pure_lo = tuple(l + 24 for l in lo)
    a = big[(slice(None),) + shard.local(pure_lo, hi, shard.accum_origin)]
    b = small[(slice(None),) + shard.local(pure_lo, hi, lo)]
    assert a.shape == b.shape and a.numel() > 0
    assert bool((b != 0).any())
    assert torch.equal(a, b)
    # and the low corner (smallest indices), against its own small block
    zs, ys, xs = (sorted({s[i] for s in shard.starts})[:2] for i in range(3))
    sub = [(z, y, x) for z in zs for y in ys for x in xs]
    lo = (zs[0], ys[0], xs[0])
    dims = (zs[-1] + 96 - lo[0], ys[-1] + 96 - lo[1], xs[-1] + 96 - lo[2])
    svol, sblk = block_volume(lo, dims)
    small = inference.run_sliding_window(svol, model, plan, 3, 16, 1000, mn, mx, starts=sub,
                                         accum_block=sblk)
    # up to where the third patch along an axis starts writing (start + 128 + 8)
    pure_hi = tuple(l + 136 for l in lo)
    a = big[(slice(None),) + shard.local(lo, pure_hi, shard.accum_origin)]
    b = small[(slice(None),) + shard.local(lo, pure_hi, lo)]
    assert torch.equal(a, b) and bool((b != 0).any())
    # percentiles over 2.28e9 voxels (64-bit counts) and the final division at these sizes
    p1, p999 = inference.volume_percentiles(volume, 1000, (1, 99.9))
    assert 18.0 <= p1 <= 21.0 and p999 == 1000.0
    inference.stitch_finalize(big, plan, blk)
    inference.stitch_finalize(small, plan, sblk)
    a = big[(slice(None),) + shard.local(lo, pure_hi, shard.accum_origin)]
    assert torch.equal(a, small[(slice(None),) + shard.local(lo, pure_hi, lo)])
    assert float(a.max()) <= 1.0 and float(a[:, 8:, 8:, 8:].min()) > 0.0


def _stream_worker(rank, world, port, failures):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from aind_exaspim_neuron_segmentation_amd import inference, sharding

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        model = _model(dev)
        gvol = synthetic.synth_volume(GSHAPE, seed=3)
        plan = inference.SlidingWindow(GSHAPE, KW["patch_shape"], KW["overlap"], KW["trim"])
        shard = sharding.Shard(plan, sharding.rank_grid(world), rank)
        group = dist.group.WORLD
        # the whole-accumulator route
        core_sl = tuple(slice(o, o + d) for o, d in zip(shard.core_origin, shard.core_dims))
        core = torch.from_numpy(np.ascontiguousarray(gvol[core_sl]).view(np.int16)).to(dev)
        block = sharding.exchange_input_halo(core, shard, group)
        volume = inference.DeviceVolume(block, np.uint16, shard.input_origin, GSHAPE)
        accum = sharding.predict_shard(volume, model, plan, shard, n_channels=3, batch_size=5, group=group)
        want = sharding.owned_result(accum, shard).cpu().numpy()
        # reader -> rank block -> result, resident input
        got = sharding.predict_shard_streaming(gvol, model, plan, shard, batch_size=5, group=group)
        np.testing.assert_array_equal(got, want)
        # read_box function, input read slab by slab, sink, thin slabs, half-precision export
        inference.release_pinned_buffers()
        old_cap = inference.PINNED_SLOT_BYTES
        inference.PINNED_SLOT_BYTES = 5 * 3 * GSHAPE[1] * GSHAPE[2] * 4      # five planes of the global plane
        try:
            boxes = []

            def read_box(lo, hi):
                in_hi = tuple(o + d for o, d in zip(shard.input_origin, shard.input_dims))
                assert all(a >= o and b <= h for a, b, o, h in zip(lo, hi, shard.input_origin, in_hi))
                return gvol[tuple(slice(a, b) for a, b in zip(lo, hi))]

            def sink(lo, hi, blk):
                assert blk.dtype == np.float16 and hi[0] - lo[0] <= 5
                boxes.append((lo, hi, blk.copy()))

            out = sharding.predict_shard_streaming(read_box, model, plan, shard, batch_size=5, group=group,
                                                   dtype=np.uint16, keep_input_resident=False,
                                                   write_block=sink, out_dtype=np.float16)
        finally:
            inference.PINNED_SLOT_BYTES = old_cap
        assert out is None
        full = np.full(want.shape, np.nan, np.float16)
        for lo, hi, blk in boxes:
            full[(slice(None),) + tuple(slice(a - o, b - o) for a, b, o in zip(lo, hi, shard.own_lo))] = blk
        np.testing.assert_array_equal(full.view(np.uint16), want.astype(np.float16).view(np.uint16))
    except Exception as exc:
        failures.put(f"rank {rank}: {type(exc).__name__}: {exc}")
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_streamed_shards_equal_predict_shard_bit_for_bit(world):
    """sharding.predict_shard_streaming (reader -> rank blocks -> sink) with the real
    kernels: each rank reads only its block of the global array, parks its first z-band planes until
    the -z neighbour's band arrives, trades y rows slab by slab -- and its output region equals
    predict_shard's (whole block resident, whole accumulator returned) bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    ctx = mp.get_context("spawn")
    failures = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, world, port, failures)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    msgs = []
    while not failures.empty():
        msgs.append(failures.get())
    assert not msgs, msgs
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def test_streamed_shard_footprint_does_not_grow_with_the_block_depth():
    """One rank of a z-split grid, run on its own (group=None: its bands simply stay where they are):
    with a read_box source, a sink and keep_input_resident=False the device holds one input slab,
    two one-layer accumulators, the parked band planes and the output slots whatever the depth of
    the rank's block. A 4x deeper block peaks at the same device memory."""
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from aind_exaspim_neuron_segmentation_amd import inference, sharding

    dev = torch.device("cuda:0")
    model = _model(dev)
    kw = dict(patch_shape=(32, 32, 32), overlap=(8, 8, 8), trim=4)

    def run(depth):
        gshape = (2 * depth, 96, 96)
        gvol = synthetic.synth_volume(gshape, seed=3)
        plan = inference.SlidingWindow(gshape, kw["patch_shape"], kw["overlap"], kw["trim"])
        shard = sharding.Shard(plan, (2, 1), 1)
        csum = [0.0]
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_allocated(dev)
        sharding.predict_shard_streaming(lambda lo, hi: gvol[tuple(slice(a, b) for a, b in zip(lo, hi))],
                                         model, plan, shard, batch_size=9, dtype=gvol.dtype,
                                         keep_input_resident=False,
                                         write_block=lambda lo, hi, b: csum.__setitem__(0, csum[0] + float(b.sum())))
        assert csum[0] > 0
        return torch.cuda.max_memory_allocated(dev) - base

    run(104)        # warm-up: workspace and staging buffers exist afterwards
    shallow, deep = run(264), run(1056)
    print(f"peak device bytes above baseline: block depth 264 -> {shallow}, 1056 -> {deep}")
    assert deep <= shallow + (1 << 20)
