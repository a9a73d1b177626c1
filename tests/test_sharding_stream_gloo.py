"""
sharding.predict_shard_streaming on CPU (gloo, world sizes 1, 2, 4 and 8 = the 4 x 2 grid):
reader -> rank blocks -> sink. The device steps (upload, histogram, gather + network + stitch
of a patch layer, the final division, the download pipeline) are replaced by numpy restatements
through the function's own `ops` hook; the schedule under test is the product's: which boxes
of the image a rank reads, the rolling input slab, the one-layer accumulators, the parked
z-band planes, the per-slab y exchange, the z exchange at the end, the boxes handed to the sink.

The assembled result must equal predict_shard's whole-accumulator route (accumulate everything,
exchange_output_bands, divide) BIT FOR BIT, with float patch outputs whose sums depend on the
order of the additions.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aind_exaspim_neuron_segmentation_amd import inference, sharding
from aind_exaspim_neuron_segmentation_amd.utils import img_util, synthetic

PATCH, TRIM, CHANNELS = (32, 32, 32), 4, 3
GEOMETRIES = {
    # overlap 16: stride 16, 8-voxel bands cross every rank face (and corners are forwarded)
    "bands": ((104, 88, 56), (16, 16, 16)),
    # overlap 8: trimmed outputs tile exactly, nothing crosses a face (empty band boxes)
    "tiling": ((104, 88, 56), (8, 8, 8)),
    # overlap 4 < trim... not allowed by the window; overlap 24: band 16 = two layers deep
    "deep": ((136, 72, 40), (24, 24, 24)),
}


def fake_patch_output(start):
    """Deterministic float stand-in for sigmoid(model(patch)), keyed by the patch start."""
    seed = (start[0] * 1000 + start[1]) * 1000 + start[2]
    u = synthetic._uniform01("patch", CHANNELS * int(np.prod(PATCH)), seed)
    return u.astype(np.float32).reshape((CHANNELS,) + PATCH)


def add_patch(accum, origin, start, gshape):
    """numpy restatement of exaspim_stitch_accumulate for one patch on a block at "origin"."""
    out = [p - 2 * TRIM for p in PATCH]
    pred = fake_patch_output(start)
    s0 = [si + TRIM for si in start]
    e = [min(a + o, d) for a, o, d in zip(s0, out, gshape)]
    dst = tuple(slice(a - o, b - o) for a, b, o in zip(s0, e, origin))
    src = tuple(slice(TRIM, TRIM + b - a) for a, b in zip(s0, e))
    accum[(slice(None),) + dst] += pred[(slice(None),) + src]


def weights(plan):
    wgt = np.zeros(plan.shape, np.float32)
    out = [p - 2 * TRIM for p in PATCH]
    for s in plan.starts():
        s0 = [si + TRIM for si in s]
        e = [min(a + o, d) for a, o, d in zip(s0, out, plan.shape)]
        wgt[tuple(slice(a, b) for a, b in zip(s0, e))] += 1
    return wgt


class SyncDrain:
    """The download pipeline without a device: fill, then run the consumers at once."""

    def emit(self, shape, fill, consumers):
        out = torch.zeros(shape, dtype=torch.float32)
        fill(out)
        for job in consumers(out.numpy()):
            job()

    def drain(self):
        pass

    def close(self):
        pass


class NumpyShardOps:
    """numpy restatements of DeviceShardOps' steps; also checks what the schedule feeds them."""

    device = torch.device("cpu")

    def __init__(self, plan, shard, gvol, slab_planes):
        self.plan, self.shard, self.gvol = plan, shard, gvol
        self.wgt = weights(plan)
        self.slab_planes = slab_planes
        self.read_voxels = 0

    def storage(self, src_dtype):
        assert np.dtype(src_dtype) == np.uint16
        return np.dtype(np.uint16), lambda block: block

    def upload(self, block, convert):
        self.read_voxels += int(block.size)
        return torch.from_numpy(np.ascontiguousarray(convert(block)).astype(np.int32))

    def empty_voxels(self, dims, vdtype):
        return torch.full(tuple(dims), -1, dtype=torch.int32)

    def effective_clip(self, src_dtype, vdtype):
        return np.uint16(1000), np.dtype(np.uint16)

    def histogram_into(self, hist, voxels, vdtype, clip, pass_index, prefix):
        v = np.minimum(voxels.numpy(), int(clip)).ravel()
        assert v.min() >= 0
        hist += torch.from_numpy(np.bincount(v, minlength=65536))

    def percentiles(self, histogram, vdtype, percentiles, value_dtype, clip):
        stats = img_util.OrderStatistics(histogram(), lambda b: np.uint16(b))
        return img_util.percentiles_from_statistics(stats, percentiles, np.uint16)

    def run_layer(self, voxels, vox_origin, src_dtype, vdtype, starts, accum, accum_origin, mn, mx, pbar=None):
        g = self.plan.shape
        for s in starts:
            # every voxel the patch reads lies inside what the schedule put on the "device", and
            # holds the global volume's value there
            lo = tuple(s)
            hi = tuple(min(a + p, d) for a, p, d in zip(s, PATCH, g))
            loc = tuple(slice(a - o, b - o) for a, b, o in zip(lo, hi, vox_origin))
            assert all(sl.start >= 0 and sl.stop <= n for sl, n in zip(loc, voxels.shape)), (s, vox_origin)
            np.testing.assert_array_equal(voxels[loc].numpy(), self.gvol[tuple(slice(a, b) for a, b in zip(lo, hi))])
            add_patch(accum.numpy(), accum_origin, s, g)

    def finalize(self, out, origin):
        sl = tuple(slice(o, o + n) for o, n in zip(origin, out.shape[1:]))
        w = self.wgt[sl]
        np.divide(out.numpy(), w, out=out.numpy(), where=w != 0)

    def zeros(self, shape):
        return torch.zeros(tuple(shape), dtype=torch.float32)

    def empty(self, shape):
        return torch.full(tuple(shape), float("nan"), dtype=torch.float32)

    def slab_bytes_cap(self):
        g = self.plan.shape
        return self.slab_planes * CHANNELS * g[1] * g[2] * 4

    def make_drain(self, slot_elems, threads):
        return SyncDrain()

    def synchronize(self):
        pass


def reference_route(plan, shard, group):
    """predict_shard's route on the CPU: whole accumulator, exchange_output_bands, divide."""
    accum = np.zeros((CHANNELS,) + shard.accum_dims, np.float32)
    for s in shard.starts:
        add_patch(accum, shard.accum_origin, s, plan.shape)
    accum_t = torch.from_numpy(accum)
    if group is not None:
        sharding.exchange_output_bands(accum_t, shard, group)
    own = sharding.owned_result(accum_t, shard).numpy().copy()
    w = weights(plan)[tuple(slice(a, b) for a, b in zip(shard.own_lo, shard.own_hi))]
    np.divide(own, w, out=own, where=w != 0)
    return own


def run_rank(rank, world, geometry, resident, slab_planes):
    gshape, overlap = GEOMETRIES[geometry]
    group = dist.group.WORLD if world > 1 else None
    plan = inference.SlidingWindow(gshape, PATCH, overlap, TRIM)
    shard = sharding.Shard(plan, sharding.rank_grid(world), rank)
    gvol = synthetic.synth_volume(gshape, seed=3)
    want = reference_route(plan, shard, group)

    # (a) array-like source, whole region returned
    ops = NumpyShardOps(plan, shard, gvol, slab_planes)
    got = sharding.predict_shard_streaming(gvol, None, plan, shard, group=group, ops=ops,
                                           keep_input_resident=resident)
    assert got.shape == want.shape and got.dtype == np.float32
    np.testing.assert_array_equal(got, want)
    # the rank read its own block only: sub-volume once for the histogram + block once for the
    # patches (resident: the block once)
    core, block = int(np.prod(shard.core_dims)), int(np.prod(shard.input_dims))
    assert ops.read_voxels <= (block if resident else core + block * 2), (ops.read_voxels, core, block)

    # (b) read_box function + write_block sink: boxes tile the region exactly once
    requested = []

    def read_box(lo, hi):
        requested.append((lo, hi))
        in_lo = shard.input_origin
        in_hi = tuple(o + d for o, d in zip(in_lo, shard.input_dims))
        assert all(a >= l and b <= h for a, b, l, h in zip(lo, hi, in_lo, in_hi)), (lo, hi, in_lo, in_hi)
        return gvol[tuple(slice(a, b) for a, b in zip(lo, hi))]

    cover = np.zeros(want.shape[1:], np.int32)
    parts = np.full(want.shape, np.nan, np.float32)

    def sink(lo, hi, block):
        sl = tuple(slice(a - o, b - o) for a, b, o in zip(lo, hi, shard.own_lo))
        assert block.shape == (CHANNELS,) + tuple(b - a for a, b in zip(lo, hi))
        cover[sl] += 1
        parts[(slice(None),) + sl] = block

    ops2 = NumpyShardOps(plan, shard, gvol, slab_planes)
    out = sharding.predict_shard_streaming(read_box, None, plan, shard, group=group, ops=ops2,
                                           dtype=np.uint16, write_block=sink, keep_input_resident=resident)
    assert out is None and requested
    assert cover.min() == 1 and cover.max() == 1
    np.testing.assert_array_equal(parts, want)
    return shard, got


def _worker(rank, world, port, failures, geometry, resident, slab_planes):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        shard, got = run_rank(rank, world, geometry, resident, slab_planes)
        # the ranks' regions assemble to the single-process overlap-add (order of the band sums
        # differs from one process at rank faces: 1 ulp)
        gshape, overlap = GEOMETRIES[geometry]
        parts = [None] * world
        dist.gather_object((shard.own_lo, shard.own_hi, got), parts if rank == 0 else None, dst=0)
        if rank == 0:
            plan = inference.SlidingWindow(gshape, PATCH, overlap, TRIM)
            full = np.full((CHANNELS,) + gshape, np.nan, np.float32)
            for lo, hi, arr in parts:
                full[(slice(None),) + tuple(slice(a, b) for a, b in zip(lo, hi))] = arr
            assert not np.isnan(full).any()
            one = sharding.Shard(plan, (1, 1), 0)
            np.testing.assert_allclose(full, reference_route(plan, one, None), rtol=0, atol=2e-6)
    except Exception as exc:  # surface the failure in the parent
        failures.put(f"rank {rank}: {type(exc).__name__}: {exc}")
        raise
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,geometry,resident,slab_planes", [
    (2, "bands", False, 5), (4, "bands", True, 3), (8, "bands", False, 7), (8, "deep", False, 4),
    (4, "tiling", False, 6), (2, "deep", True, 100),
])
def test_streamed_shards_equal_the_whole_accumulator_route(world, geometry, resident, slab_planes):
    ctx = mp.get_context("spawn")
    failures = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, failures, geometry, resident, slab_planes))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    msgs = []
    while not failures.empty():
        msgs.append(failures.get())
    assert not msgs, msgs
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


@pytest.mark.parametrize("geometry", ["bands", "tiling", "deep"])
def test_single_rank_streaming_needs_no_group(geometry):
    """World size 1: no neighbours, nothing parked, no exchange; the same code path as the ranks'."""
    run_rank(0, 1, geometry, False, 3)
