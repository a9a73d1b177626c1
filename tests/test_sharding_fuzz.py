"""
Seeded fuzz of the sharding layer's geometry on CPU (no process group: the ranks
run as threads and sharding._p2p is replaced by in-memory mailboxes, so the REAL
exchange_input_halo / exchange_output_bands / owned_result code runs for every
rank of 2-, 4- and 8-rank grids over ~150 random volume / patch / overlap / trim
combinations). Patch outputs are small integers, so sums are exact whatever the
order: after the band exchange every rank's owned region must equal the
single-process overlap-add, the owned regions must tile the volume, and every
assembled input block must equal the global volume's.
"""

import queue
import threading

import numpy as np
import pytest
import torch

from aind_exaspim_neuron_segmentation_amd import inference, sharding


class _Group:
    def __init__(self, rank, mail):
        self.rank, self.mail = rank, mail


def _mailbox_p2p(ops, group):
    for kind, tensor, peer in ops:
        if kind == "send":
            group.mail[(group.rank, peer)].put(tensor.clone())
    for kind, tensor, peer in ops:
        if kind == "recv":
            tensor.copy_(group.mail[(peer, group.rank)].get(timeout=60))


def _patch_values(plan, start, lo, hi):
    """Integer-valued stand-in for the trimmed network output of the patch at "start" on
    the global box [lo, hi): depends on the patch and on the voxel."""
    zz, yy, xx = np.meshgrid(*(np.arange(a, b) for a, b in zip(lo, hi)), indexing="ij")
    return ((zz * 7 + yy * 3 + xx + start[0] * 5 + start[1] * 11 + start[2] * 13) % 17).astype(np.float32)


def _accumulate(plan, accum, origin, starts):
    g, p, t = plan.shape, plan.patch_shape, plan.trim
    for s in starts:
        lo = tuple(a + t for a in s)
        hi = tuple(min(a + ps - 2 * t, d) for a, ps, d in zip(lo, p, g))
        if any(h <= l for l, h in zip(lo, hi)):
            continue
        dst = tuple(slice(a - o, b - o) for a, b, o in zip(lo, hi, origin))
        accum[(0,) + dst] += _patch_values(plan, s, lo, hi)


def _geometries(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        patch = tuple(int(rng.integers(6, 25)) for _ in range(3))
        trim = int(rng.integers(0, (min(patch) - 1) // 2))
        overlap = tuple(int(rng.integers(0, p - 1)) for p in patch)
        world = int(rng.choice([2, 4, 8]))
        gz, gy = sharding.rank_grid(world)
        # enough starts along z and y for the grid, a few more at random
        shape = []
        for axis, (p, o) in enumerate(zip(patch, overlap)):
            need = (gz, gy, 1)[axis]
            nstarts = need + int(rng.integers(0, 4))
            shape.append(o + (p - o) * nstarts - int(rng.integers(0, p - o)))
        out.append(dict(shape=tuple(shape), patch=patch, overlap=overlap, trim=trim, world=world))
    return out


@pytest.mark.parametrize("geo", _geometries(150, seed=11),
                         ids=lambda g: f"w{g['world']}-" + "x".join(map(str, g["shape"])))
def test_sharded_geometry_fuzz(geo, monkeypatch):
    try:
        plan = inference.SlidingWindow(geo["shape"], geo["patch"], geo["overlap"], geo["trim"])
    except ValueError as exc:           # the reference itself raises on this geometry
        assert "broadcast" in str(exc)
        return
    world = geo["world"]
    grid = sharding.rank_grid(world)
    try:
        shards = [sharding.Shard(plan, grid, r) for r in range(world)]
    except ValueError as exc:           # refused loudly: too few starts or a band wider than a rank
        assert "cannot be split" in str(exc) or "overlap band" in str(exc)
        return
    monkeypatch.setattr(sharding, "_p2p", _mailbox_p2p)
    gshape = plan.shape
    gvol = (np.arange(int(np.prod(gshape)), dtype=np.int64) * 2654435761 % 30011).astype(np.int16).reshape(gshape)
    want = np.zeros((1,) + gshape, np.float32)
    _accumulate(plan, want, (0, 0, 0), plan.starts())

    # 1. the ranks partition the patch list
    every = [s for sh in shards for s in sh.starts]
    assert sorted(every) == sorted(plan.starts()) and len(set(every)) == len(every)

    mail = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    owned = np.zeros(gshape, np.int32)
    errors = []

    def run(rank):
        try:
            sh = shards[rank]
            group = _Group(rank, mail)
            core_sl = tuple(slice(o, o + d) for o, d in zip(sh.core_origin, sh.core_dims))
            core = torch.from_numpy(np.ascontiguousarray(gvol[core_sl]))
            # 2. input block = sub-volume + halo from the neighbours
            block = sharding.exchange_input_halo(core, sh, group).numpy()
            in_sl = tuple(slice(o, o + d) for o, d in zip(sh.input_origin, sh.input_dims))
            np.testing.assert_array_equal(block, gvol[in_sl])
            # 3. overlap-add of my patches, bands to their owners
            accum = np.zeros((1,) + sh.accum_dims, np.float32)
            _accumulate(plan, accum, sh.accum_origin, sh.starts)
            accum = torch.from_numpy(accum)
            sharding.exchange_output_bands(accum, sh, group)
            own = sharding.owned_result(accum, sh).numpy()
            own_sl = tuple(slice(a, b) for a, b in zip(sh.own_lo, sh.own_hi))
            np.testing.assert_array_equal(own, want[(slice(None),) + own_sl])
            owned[own_sl] += 1          # disjoint slices: no two threads touch the same voxel
        except Exception as exc:        # noqa: BLE001 - reported by the main thread
            errors.append(f"rank {rank}: {type(exc).__name__}: {exc}")

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, (geo, errors[:2])
    # 4. the owned regions tile the volume
    assert (owned == 1).all(), geo
