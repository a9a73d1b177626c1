"""
Multi-GPU sharding of predict()'s sliding window: one process per GPU,
``torch.distributed`` (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

The reference is single-device (inference.py:85-117). Patches are independent
units that meet only in (1) the global percentile pair and (2) the additive
overlap bands of the stitch (inference.py:115-116), so the path shards by
sub-volume on the reference's GLOBAL patch grid:

* the z and y patch-start lists are split into contiguous runs over a
  (gz, gy) rank grid; a rank processes the patches whose start it owns;
* a rank's input block is the union of its patches' in-volume extents, i.e.
  its own sub-volume plus an "overlap"-voxel halo from the next rank in z / y;
  ``exchange_input_halo`` fetches that halo from the neighbours when every
  rank holds only its disjoint sub-volume;
* its patches' trimmed outputs reach (patch - 2*trim - stride) voxels into the
  next rank's output region; after the last batch those partial-sum bands are
  sent to the owner and added (``exchange_output_bands``): first along z, then
  along y, so corner contributions are forwarded;
* the percentile histogram is summed over ranks (512 KiB all-reduce).

No other communication happens; the network never sees a neighbour's data.
Sums of the overlap bands are associated differently from the single-device
order at rank faces (float addition: differences of one ulp, far below the
1e-3 parity bar); everything else is identical.
"""

import numpy as np
import torch

from aind_exaspim_neuron_segmentation_amd import _native


def rank_grid(world):
    """
    (gz, gy) rank grid for a world size: gy is the largest divisor of the world
    size with gy * gy <= world, z takes the rest
    (1 -> 1x1, 2 -> 2x1, 4 -> 2x2, 8 -> 4x2).
    """
    gy = max(d for d in range(1, world + 1) if world % d == 0 and d * d <= world)
    return world // gy, gy


def _split(n, parts, index):
    """Contiguous, near-even split of range(n) into parts; returns (lo, hi)."""
    base, extra = divmod(n, parts)
    lo = index * base + min(index, extra)
    return lo, lo + base + (1 if index < extra else 0)


class Shard:
    """
    The part of the sliding window one rank owns.

    Parameters
    ----------
    plan : inference.SlidingWindow
        Window geometry over the GLOBAL volume.
    grid : Tuple[int]
        (gz, gy) rank grid.
    rank : int
        Rank of this process (row-major over the grid: rank = iz * gy + iy).

    Attributes
    ----------
    starts : List[Tuple[int]]
        Global patch starts this rank processes (reference order).
    input_origin, input_dims : Tuple[int]
        Block of the input volume the rank's patches read.
    core_origin, core_dims : Tuple[int]
        Disjoint sub-volume of the input the rank "owns" (histogram, halo).
    own_lo, own_hi : Tuple[int]
        Disjoint output region whose final sums live on this rank (global).
    """

    def __init__(self, plan, grid, rank):
        self.plan = plan
        self.grid = tuple(grid)
        gz, gy = self.grid
        self.rank = rank
        self.iz, self.iy = divmod(rank, gy)
        g = plan.shape
        p, ov, trim = plan.patch_shape, plan.overlap, plan.trim
        axis_starts = [
            list(range(0, d - ps + (ps - o), ps - o)) for d, ps, o in zip(g, p, ov)
        ]
        self.axis_starts = axis_starts
        self.ranges = []
        for axis, (parts, idx) in enumerate(((gz, self.iz), (gy, self.iy), (1, 0))):
            if len(axis_starts[axis]) < parts:
                raise ValueError(
                    f"axis {axis}: {len(axis_starts[axis])} patch starts cannot be split {parts} ways"
                )
            self.ranges.append(_split(len(axis_starts[axis]), parts, idx))
            # exchange_output_bands hands the overlap band one rank forward along an
            # axis and no further: the band must fit into the owned extent of every
            # rank that receives one and passes its own on (the interior ranks)
            band = p[axis] - 2 * trim - (p[axis] - ov[axis])
            for j in range(1, parts - 1):
                lo_j, hi_j = _split(len(axis_starts[axis]), parts, j)
                extent = (hi_j - lo_j) * (p[axis] - ov[axis])
                if band > extent:
                    raise ValueError(
                        f"axis {axis}: the {band}-voxel overlap band does not fit into the "
                        f"{extent}-voxel region of rank {j} of {parts}; use fewer ranks along "
                        "this axis or a larger stride"
                    )
        mine = [axis_starts[a][lo:hi] for a, (lo, hi) in enumerate(self.ranges)]
        self.starts = [(z, y, x) for z in mine[0] for y in mine[1] for x in mine[2]]

        first = [m[0] for m in mine]
        last = [m[-1] for m in mine]
        parts = (gz, gy, 1)
        idx = (self.iz, self.iy, 0)
        self.input_origin = tuple(first)
        self.input_dims = tuple(min(l + ps, d) - f for f, l, ps, d in zip(first, last, p, g))
        # disjoint input sub-volume: from my first start to the next rank's first start
        core_lo, core_hi, own_lo, own_hi = [], [], [], []
        for a in range(3):
            lo_i, hi_i = self.ranges[a]
            is_first = idx[a] == 0
            is_last = idx[a] == parts[a] - 1
            nxt = g[a] if is_last else axis_starts[a][hi_i]
            core_lo.append(0 if is_first else first[a])
            core_hi.append(nxt)
            # (a last start within "trim" of the end of the volume writes nothing: clip)
            own_lo.append(0 if is_first else min(first[a] + trim, g[a]))
            own_hi.append(g[a] if is_last else min(nxt + trim, g[a]))
        self.core_origin = tuple(core_lo)
        self.core_dims = tuple(h - l for l, h in zip(core_lo, core_hi))
        self.own_lo, self.own_hi = tuple(own_lo), tuple(own_hi)
        # accumulator block = input block (covers every write of my patches); with an overlap
        # smaller than the trim the trimmed outputs of neighbouring patches leave gaps that
        # stay 0, and the gap at a rank face belongs to the owned region: stretch the block
        # over it
        self.accum_origin = self.input_origin
        self.accum_dims = tuple(
            max(o + d, h) - o for o, d, h in zip(self.input_origin, self.input_dims, self.own_hi)
        )

    # ---- neighbours -----------------------------------------------------------
    def neighbour(self, dz, dy):
        """Rank at grid offset (dz, dy), or None outside the grid."""
        z, y = self.iz + dz, self.iy + dy
        if 0 <= z < self.grid[0] and 0 <= y < self.grid[1]:
            return z * self.grid[1] + y
        return None

    def local(self, lo, hi, origin):
        """Slices of a local block (at "origin") for the global box [lo, hi)."""
        return tuple(slice(a - o, b - o) for a, b, o in zip(lo, hi, origin))

    def band_box(self, axis):
        """
        Global box of the partial sums this rank must hand to its +1 neighbour
        along "axis" (0 = z, 1 = y): everything it accumulated at or beyond the
        neighbour's output region. Along the other split axis the box spans the
        whole accumulator for z (corners are forwarded in the y phase) and the
        owned range for y.
        """
        end = [o + d for o, d in zip(self.accum_origin, self.accum_dims)]
        lo = list(self.accum_origin)
        hi = list(end)
        lo[axis] = self.own_hi[axis]
        if axis == 1:
            lo[0], hi[0] = self.own_lo[0], self.own_hi[0]
            lo[0] = max(lo[0], self.accum_origin[0])
        return tuple(lo), tuple(hi)


def _needs_host_staging(tensor, group):
    """gloo moves CPU tensors only: device tensors are staged through the host
    (CPU tests, and 2-rank rehearsals on a single GPU); nccl/RCCL moves them
    directly over xGMI."""
    import torch.distributed as dist

    return tensor.is_cuda and dist.get_backend(group) == "gloo"


_RCCL_DTYPES = (torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int8,
                torch.uint8, torch.int32, torch.int64)


def _p2p(ops, group):
    """Runs a batch of (kind, tensor, peer) point-to-point transfers."""
    import torch.distributed as dist

    if not ops:
        return
    staged, reqs = [], []
    bytewise = dist.get_backend(group) == "nccl"
    for kind, tensor, peer in ops:
        buf = tensor
        if _needs_host_staging(tensor, group):
            buf = tensor.cpu() if kind == "send" else torch.empty(
                tensor.shape, dtype=tensor.dtype, device="cpu")
            if kind == "recv":
                staged.append((tensor, buf))
        elif bytewise and tensor.dtype not in _RCCL_DTYPES:
            # RCCL has no 16-bit integer type (the voxels of an input halo): a copy is a
            # copy, the same memory travels as bytes
            buf = tensor.view(torch.uint8)
        reqs.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, buf, peer, group))
    for req in dist.batch_isend_irecv(reqs):
        req.wait()
    for dst, buf in staged:
        dst.copy_(buf)


def all_reduce_sum(tensor, group):
    """In-place sum over the ranks of a group (histogram reduction)."""
    import torch.distributed as dist

    if _needs_host_staging(tensor, group):
        host = tensor.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        tensor.copy_(host)
    else:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)


def exchange_output_bands(accum, shard, group):
    """
    Sends the overlap-band partial sums to the ranks that own them and adds the
    bands received from the -1 neighbours; z phase first, then y phase.

    Parameters
    ----------
    accum : torch.Tensor
        (C, *shard.accum_dims) partial sums of this rank (any device).
    shard : Shard
        This rank's shard.
    group : ProcessGroup
        torch.distributed group of all ranks.
    """
    import torch.distributed as dist

    for axis, (dz, dy) in enumerate(((1, 0), (0, 1))):
        nxt = shard.neighbour(dz, dy)
        prv = shard.neighbour(-dz, -dy)
        ops, recv_buf, recv_sl = [], None, None
        if nxt is not None:
            lo, hi = shard.band_box(axis)
            sl = shard.local(lo, hi, shard.accum_origin)
            send_buf = accum[(slice(None),) + sl].contiguous()
            if send_buf.numel():
                ops.append(("send", send_buf, nxt))
        if prv is not None:
            other = Shard(shard.plan, shard.grid, prv)
            lo, hi = other.band_box(axis)
            recv_sl = (slice(None),) + shard.local(lo, hi, shard.accum_origin)
            shape = (accum.shape[0],) + tuple(b - a for a, b in zip(lo, hi))
            recv_buf = torch.empty(shape, dtype=accum.dtype, device=accum.device)
            if recv_buf.numel():
                ops.append(("recv", recv_buf, prv))
        _p2p(ops, group)
        if recv_buf is not None and recv_buf.numel():
            accum[recv_sl] += recv_buf


def exchange_input_halo(core, shard, group):
    """
    Assembles a rank's input block (its disjoint sub-volume plus the halo its
    last patches read) from the ranks' disjoint sub-volumes: the halo comes
    from the +z, +y and +z+y neighbours.

    Parameters
    ----------
    core : torch.Tensor
        (*shard.core_dims) voxels of this rank's disjoint sub-volume.
    shard : Shard
        This rank's shard.
    group : ProcessGroup
        torch.distributed group of all ranks.

    Returns
    -------
    torch.Tensor
        (*shard.input_dims) block starting at shard.input_origin.
    """
    import torch.distributed as dist

    block = torch.zeros(shard.input_dims, dtype=core.dtype, device=core.device)
    in_lo = shard.input_origin
    in_hi = tuple(o + d for o, d in zip(in_lo, shard.input_dims))

    def overlap(a_lo, a_hi, b_lo, b_hi):
        lo = tuple(max(x, y) for x, y in zip(a_lo, b_lo))
        hi = tuple(min(x, y) for x, y in zip(a_hi, b_hi))
        return (lo, hi) if all(h > l for l, h in zip(lo, hi)) else None

    core_lo = shard.core_origin
    core_hi = tuple(o + d for o, d in zip(core_lo, shard.core_dims))
    box = overlap(in_lo, in_hi, core_lo, core_hi)
    block[shard.local(*box, in_lo)] = core[shard.local(*box, core_lo)]
    ops, pending = [], []
    world = shard.grid[0] * shard.grid[1]
    for peer in range(world):
        if peer == shard.rank:
            continue
        other = Shard(shard.plan, shard.grid, peer)
        o_core_lo = other.core_origin
        o_core_hi = tuple(o + d for o, d in zip(o_core_lo, other.core_dims))
        o_in_lo = other.input_origin
        o_in_hi = tuple(o + d for o, d in zip(o_in_lo, other.input_dims))
        need = overlap(in_lo, in_hi, o_core_lo, o_core_hi)  # what I read from peer's core
        give = overlap(o_in_lo, o_in_hi, core_lo, core_hi)  # what peer reads from my core
        if give is not None:
            buf = core[shard.local(*give, core_lo)].contiguous()
            ops.append(("send", buf, peer))
        if need is not None:
            shape = tuple(h - l for l, h in zip(*need))
            buf = torch.empty(shape, dtype=core.dtype, device=core.device)
            ops.append(("recv", buf, peer))
            pending.append((need, buf))
    _p2p(ops, group)
    for need, buf in pending:
        block[shard.local(*need, in_lo)] = buf
    return block


def _timed(timings, key, device, fn):
    """Runs fn(); with a timings dict, between device synchronisations, adding the wall
    seconds to timings[key] and to timings["seconds"] (the total of all exchange steps)."""
    import time

    if timings is None:
        return fn()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    timings[key] = timings.get(key, 0.0) + dt
    timings["seconds"] = timings.get("seconds", 0.0) + dt
    return out


def predict_shard(volume, model, plan, shard, n_channels=3, batch_size=16,
                  brightness_clip=1000, normalization_percentiles=(1, 99.9), group=None,
                  n_streams=1, timings=None, core=None):
    """
    Runs one rank's share of predict() on its device: global percentiles
    (histogram all-reduce), the rank's patches, the band exchange and the final
    division. Returns the finalised (C, *shard.accum_dims) device tensor; the
    voxels in [shard.own_lo, shard.own_hi) are this rank's part of the result.

    Parameters
    ----------
    volume : inference.DeviceVolume
        The rank's input block (shard.input_origin / input_dims) on its device.
    model : torch.nn.Module
        Network replica on the same device.
    plan : inference.SlidingWindow
        Window geometry over the global volume.
    shard : Shard
        This rank's shard.
    group : ProcessGroup, optional
        Process group (None = single process).
    timings : dict, optional
        If given, the wall time this rank spends in the exchange steps, measured
        between device synchronisations, is added to timings["histogram_s"] (the
        all-reduce of the percentile histograms), timings["output_bands_s"] and
        their total timings["seconds"].
    core : torch.Tensor, optional
        The rank's disjoint sub-volume (shard.core_dims, contiguous, on the
        device) when the caller holds it anyway (it fed exchange_input_halo):
        the histogram then runs over it instead of over a copy cut out of
        "volume".
    """
    from aind_exaspim_neuron_segmentation_amd import inference

    device = volume.tensor.device
    multi = group is not None and shard.grid[0] * shard.grid[1] > 1
    if multi:
        if core is None:
            core_sl = shard.local(
                shard.core_origin,
                tuple(o + d for o, d in zip(shard.core_origin, shard.core_dims)),
                shard.input_origin,
            )
            core = volume.tensor[core_sl].contiguous()
        elif tuple(core.shape) != tuple(shard.core_dims):
            raise ValueError(f"core has shape {tuple(core.shape)}, expected {tuple(shard.core_dims)}")
        core_vol = inference.DeviceVolume(
            core, volume.np_dtype, shard.core_origin, plan.shape, storage_dtype=volume.storage_dtype,
        )

        def reduce_fn(hist):
            _timed(timings, "histogram_s", device, lambda: all_reduce_sum(hist, group))

        mn, mx = inference.volume_percentiles(core_vol, brightness_clip, normalization_percentiles,
                                              reduce_fn=reduce_fn)
    else:
        mn, mx = inference.volume_percentiles(volume, brightness_clip, normalization_percentiles)

    accum_block = _native.Block.make(shard.accum_dims, shard.accum_origin, plan.shape)
    accum = inference.run_sliding_window(
        volume, model, plan, n_channels, batch_size, brightness_clip, mn, mx,
        starts=shard.starts, accum_block=accum_block, n_streams=n_streams,
    )
    if multi:
        _timed(timings, "output_bands_s", device, lambda: exchange_output_bands(accum, shard, group))
    inference.stitch_finalize(accum, plan, accum_block)
    return accum


def owned_result(accum, shard):
    """Slices a finalised accumulator down to the rank's disjoint output region."""
    sl = shard.local(shard.own_lo, shard.own_hi, shard.accum_origin)
    return accum[(slice(None),) + sl]
