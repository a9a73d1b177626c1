"""
Multi-process tests of the sharding layer on CPU (gloo, world sizes 2, 4 and 8;
one geometry whose trimmed outputs tile exactly and one with 8-voxel overlap
bands that really travel between ranks):
partition of the global patch grid, input halo exchange, histogram reduction
and output band exchange. The device kernels are replaced by their numpy
restatements so the N > 1 logic is covered without a GPU.
"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aind_exaspim_neuron_segmentation_amd import inference, sharding
from aind_exaspim_neuron_segmentation_amd.utils import synthetic

GSHAPE = (104, 88, 56)
PATCH, TRIM = (32, 32, 32), 4
CHANNELS = 2
# overlap 8: stride 24 = trimmed output size, no overlap bands (pure partition);
# overlap 16: stride 16, every trimmed output reaches 8 voxels into the next patch
OVERLAPS = {"tiling": (8, 8, 8), "bands": (16, 16, 16)}


def fake_patch_output(start):
    """Deterministic stand-in for sigmoid(model(patch)) keyed by the patch start."""
    seed = (start[0] * 1000 + start[1]) * 1000 + start[2]
    u = synthetic._uniform01("patch", CHANNELS * int(np.prod(PATCH)), seed)
    return u.astype(np.float32).reshape((CHANNELS,) + PATCH)


def accumulate(accum, origin, starts, gshape):
    """numpy restatement of exaspim_stitch_accumulate on a block at "origin"."""
    out = [p - 2 * TRIM for p in PATCH]
    for s in starts:
        pred = fake_patch_output(s)
        s0 = [si + TRIM for si in s]
        e = [min(a + o, d) for a, o, d in zip(s0, out, gshape)]
        dst = tuple(slice(a - o, b - o) for a, b, o in zip(s0, e, origin))
        src = tuple(slice(TRIM, TRIM + b - a) for a, b in zip(s0, e))
        accum[(slice(None),) + dst] += pred[(slice(None),) + src]


def expected_result(overlap):
    plan = inference.SlidingWindow(GSHAPE, PATCH, overlap, TRIM)
    accum = np.zeros((CHANNELS,) + GSHAPE, np.float32)
    accumulate(accum, (0, 0, 0), plan.starts(), GSHAPE)
    wgt = np.zeros((1,) + GSHAPE, np.float32)
    out = [p - 2 * TRIM for p in PATCH]
    for s in plan.starts():
        s0 = [si + TRIM for si in s]
        e = [min(a + o, d) for a, o, d in zip(s0, out, GSHAPE)]
        wgt[(0,) + tuple(slice(a, b) for a, b in zip(s0, e))] += 1
    return accum, wgt


def _worker(rank, world, port, failures, overlap):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        group = dist.group.WORLD
        plan = inference.SlidingWindow(GSHAPE, PATCH, overlap, TRIM)
        grid = sharding.rank_grid(world)
        shard = sharding.Shard(plan, grid, rank)
        gvol = synthetic.synth_volume(GSHAPE, seed=3)

        # 1. the shards partition the global patch list
        mine = torch.tensor(shard.starts, dtype=torch.int64)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([len(shard.starts)]))
        assert sum(int(c) for c in counts) == len(plan.starts())
        assert set(shard.starts) <= set(plan.starts())

        # 2. input halo exchange rebuilds the block every patch of the rank reads
        core_sl = tuple(slice(o, o + d) for o, d in zip(shard.core_origin, shard.core_dims))
        core = torch.from_numpy(gvol[core_sl].astype(np.int32))
        block = sharding.exchange_input_halo(core, shard, group)
        in_sl = tuple(slice(o, o + d) for o, d in zip(shard.input_origin, shard.input_dims))
        np.testing.assert_array_equal(block.numpy(), gvol[in_sl].astype(np.int32))

        # 3. summed per-core histograms give the global percentiles
        hist = torch.from_numpy(
            np.bincount(np.minimum(gvol[core_sl], 1000).ravel(), minlength=65536)
        )
        dist.all_reduce(hist)
        from aind_exaspim_neuron_segmentation_amd.utils import img_util

        stats = img_util.OrderStatistics(hist.numpy(), lambda b: np.uint16(b))
        got = img_util.percentiles_from_statistics(stats, (1, 99.9), np.uint16)
        np.testing.assert_array_equal(got, np.percentile(np.minimum(gvol, 1000), (1, 99.9)))

        # 4. band exchange: owned region == single-process result
        accum = np.zeros((CHANNELS,) + shard.accum_dims, np.float32)
        accumulate(accum, shard.accum_origin, shard.starts, GSHAPE)
        accum_t = torch.from_numpy(accum)
        if overlap[0] > 8 and shard.neighbour(1, 0) is not None:
            lo, hi = shard.band_box(0)
            assert hi[0] - lo[0] >= 8        # a real band leaves this rank
        sharding.exchange_output_bands(accum_t, shard, group)
        want, wgt = expected_result(overlap)
        own = tuple(slice(a, b) for a, b in zip(shard.own_lo, shard.own_hi))
        got_own = sharding.owned_result(accum_t, shard).numpy()
        np.testing.assert_allclose(got_own, want[(slice(None),) + own], rtol=0, atol=2e-6)
        covered = wgt[(slice(None),) + own] > 0
        assert (got_own[:, covered[0]] > 0).all()

        # 5. the owned regions tile the volume exactly once
        cover = torch.zeros(GSHAPE, dtype=torch.int32)
        cover[own] = 1
        dist.all_reduce(cover)
        assert int(cover.min()) == 1 and int(cover.max()) == 1
    except Exception as exc:  # surface the failure in the parent
        failures.put(f"rank {rank}: {type(exc).__name__}: {exc}")
        raise
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,geometry", [(2, "tiling"), (4, "tiling"), (2, "bands"), (4, "bands"),
                                            (8, "bands")])
def test_sharded_pipeline_gloo(world, geometry):
    """world 8 = the 4 x 2 grid of an 8-GPU node: interior z ranks send and receive
    in the same phase, corner sums are forwarded z first, then y."""
    ctx = mp.get_context("spawn")
    failures = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, failures, OVERLAPS[geometry]))
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


def test_rank_grid_and_shard_geometry():
    assert [sharding.rank_grid(n) for n in (1, 2, 4, 8)] == [(1, 1), (2, 1), (2, 2), (4, 2)]
    # BASELINE config 4: 2048 x 1024 x 1024, 2-way along z
    plan = inference.SlidingWindow((2048, 1024, 1024), (96,) * 3, (32,) * 3, 8)
    s0, s1 = sharding.Shard(plan, (2, 1), 0), sharding.Shard(plan, (2, 1), 1)
    assert len(s0.starts) == len(s1.starts) == 16 * 16 * 16
    assert s0.input_origin == (0, 0, 0) and s0.input_dims == (1024 + 32, 1024, 1024)
    assert s1.input_origin == (1024, 0, 0) and s1.input_dims == (1024, 1024, 1024)
    assert s0.own_hi[0] == s1.own_lo[0] == 1024 + 8
    lo, hi = s0.band_box(0)
    assert (lo[0], hi[0]) == (1032, 1056)  # 16 voxels of sums + 8 trimmed (zero) voxels
    # BASELINE config 5: 4096 x 2048 x 2048 over a 4 x 2 grid
    plan = inference.SlidingWindow((4096, 2048, 2048), (96,) * 3, (32,) * 3, 8)
    shards = [sharding.Shard(plan, (4, 2), r) for r in range(8)]
    assert sum(len(s.starts) for s in shards) == 64 * 32 * 32
    assert all(len(s.starts) == 8192 for s in shards)
    with pytest.raises(ValueError):
        sharding.Shard(inference.SlidingWindow((96,) * 3, (96,) * 3, (32,) * 3, 8), (2, 1), 0)


def test_shard_rejects_bands_wider_than_the_next_ranks_region():
    """exchange_output_bands hands a band one rank forward per axis; a band wider than
    the next rank's owned extent would have to travel two ranks. Such geometries are
    refused (the defaults, band 16 < stride 64, are far from it)."""
    plan = inference.SlidingWindow((200, 120, 64), (64,) * 3, (48,) * 3, 4)   # band 40, stride 16
    with pytest.raises(ValueError, match="overlap band"):
        sharding.Shard(plan, (4, 2), 0)
    sharding.Shard(plan, (1, 1), 0)                      # a single rank has no bands
    # 13 starts over 4 ranks -> 3 starts x 16 = 48 >= 40 on every interior rank
    ok = inference.SlidingWindow((264, 120, 64), (64,) * 3, (48,) * 3, 4)
    sharding.Shard(ok, (4, 1), 1)
