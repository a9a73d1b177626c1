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


# --- reader -> shards -> sink: every rank streams its own block ---
class _BoxSource:
    """read_box(lo, hi) over anything that slices like a 3-D numpy array (ndarray, memmap,
    a zarr / N5 / TIFF-backed array as the reference's img_util.read returns, img_util.py:25-121)."""

    def __init__(self, arr):
        while len(arr.shape) > 3:
            if arr.shape[0] != 1:
                raise ValueError("leading image axes must have length 1")
            arr = arr[0]
        if len(arr.shape) != 3:
            raise ValueError(f"expected a 3-D image, got shape {tuple(arr.shape)}")
        self.arr = arr
        self.shape = tuple(int(v) for v in arr.shape)
        self.dtype = np.dtype(arr.dtype)

    def __call__(self, lo, hi):
        return np.asarray(self.arr[tuple(slice(a, b) for a, b in zip(lo, hi))])


class DeviceShardOps:
    """
    What predict_shard_streaming does on the device, as overridable steps (the multi-process
    CPU tests replace the kernels by their numpy restatements and keep the schedule, the band
    bookkeeping and the exchanges).
    """

    def __init__(self, model, plan, n_channels, batch_size, brightness_clip, n_streams, half_out,
                 copy_threads):
        from aind_exaspim_neuron_segmentation_amd import inference

        self.inf = inference
        self.model, self.plan = model, plan
        self.n_channels, self.batch_size = n_channels, batch_size
        self.brightness_clip, self.n_streams = brightness_clip, n_streams
        self.half_out, self.copy_threads = half_out, copy_threads
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError(
                f"predict (MI355X) has no CPU path: the model must be on a HIP device, got {self.device}")

    # -- voxels
    def storage(self, src_dtype):
        """(voxel dtype in device memory, host conversion) for an image dtype."""
        vdtype, convert = self.inf._device_voxel_dtype(src_dtype, clip=self.brightness_clip)
        if vdtype not in self.inf._VOX_CODES:
            raise TypeError(self.inf._unsupported(src_dtype))
        return vdtype, convert

    def upload(self, block, convert):
        return self.inf._carrier(convert(block)).to(self.device, non_blocking=True)

    def empty_voxels(self, dims, vdtype):
        return torch.empty(tuple(dims), dtype=self.inf._TORCH_VOXELS[np.dtype(vdtype)], device=self.device)

    def effective_clip(self, src_dtype, vdtype):
        return self.inf._effective_clip(src_dtype, self.brightness_clip, vdtype)

    def histogram_into(self, hist, voxels, vdtype, clip, pass_index, prefix):
        self.inf._histogram_into(hist, voxels.contiguous(), self.inf._VOX_CODES[np.dtype(vdtype)], clip,
                                 pass_index, prefix)

    def percentiles(self, histogram, vdtype, percentiles, value_dtype, clip):
        return self.inf._percentiles_from_histograms(histogram, vdtype, percentiles, value_dtype, clip)

    # -- one patch layer: gather -> network -> sigmoid -> trimmed overlap-add into "accum"
    def run_layer(self, voxels, vox_origin, src_dtype, vdtype, starts, accum, accum_origin, mn, mx, pbar=None):
        gshape = self.plan.shape
        volume = self.inf.DeviceVolume(voxels, src_dtype, vox_origin, gshape, storage_dtype=vdtype)
        blk = _native.Block.make(tuple(accum.shape[1:]), accum_origin, gshape)
        self.inf.run_sliding_window(volume, self.model, self.plan, self.n_channels, self.batch_size,
                                    self.brightness_clip, mn, mx, starts=starts, accum=accum,
                                    accum_block=blk, n_streams=self.n_streams, pbar=pbar)

    def finalize(self, out, origin):
        """Divides the sums of the box at global "origin" by the per-voxel patch count."""
        self.inf.stitch_finalize(out, self.plan, _native.Block.make(tuple(out.shape[1:]), origin, self.plan.shape))

    def zeros(self, shape):
        return torch.zeros(tuple(shape), dtype=torch.float32, device=self.device)

    def empty(self, shape):
        return torch.empty(tuple(shape), dtype=torch.float32, device=self.device)

    def slab_bytes_cap(self):
        return self.inf.PINNED_SLOT_BYTES

    def make_drain(self, slot_elems, threads):
        return self.inf._SlabDrain(self.device, slot_elems, self.half_out, threads)

    def synchronize(self):
        torch.cuda.synchronize(self.device)


def predict_shard_streaming(source, model, plan, shard, affinity_mode=True, batch_size=16,
                            brightness_clip=1000, normalization_percentiles=(1, 99.9), group=None,
                            write_block=None, dtype=None, keep_input_resident=None, n_streams=None,
                            copy_threads=4, out_dtype=np.float32, verbose=False, timings=None, ops=None):
    """
    One rank's share of predict() with the volume READ and the result WRITTEN in pieces: the
    rank reads only its own block of the image (its disjoint sub-volume for the percentile
    histogram, sub-volume + halo for its patches), never holds more than one patch layer of
    input, two one-layer accumulators and a few output slabs on the device, and hands every
    finished piece of its disjoint output region to "write_block" while later layers compute.
    The assembled result equals predict_shard()'s bit for bit (same additions in the same order).

    The reference needs the whole image as one array (inference.py:79); its readers
    (img_util.py:25-121) return lazily chunked zarr / N5 / TIFF arrays that are sliced on
    demand -- the interface "source" takes.

    Schedule. Pass 1: z-chunks of the rank's sub-volume -> device histogram -> all-reduce ->
    np.percentile of the clipped volume, exactly. Pass 2: the rank's patch layers in z order, as
    predict_streaming does for one device (rolling input slab, one-layer accumulator continuing
    the previous layer's overlap band). A finished z-slab of partial sums first trades its
    y-overlap rows with the +-y neighbours (same z-range on both sides, so the ranks of a grid
    row walk the same slabs), is divided and leaves through the download pipeline. The rank's
    first planes -- the z-overlap band the -z neighbour's LAST layer still adds to -- are parked
    on the device (band x plane floats) until that band arrives at the end, then take the same
    route; the rank's own last planes beyond its region are what it sends to +z. Nothing else is
    exchanged, and the device footprint does not grow with the depth of the block.

    Parameters
    ----------
    source : array-like or Callable[[Tuple[int], Tuple[int]], numpy.ndarray]
        The GLOBAL image: anything that slices like a 3-D (or 1 x 1 x D x H x W) numpy array, or
        a function read_box(lo, hi) returning voxels [lo, hi) as an array of shape hi - lo, in
        which case "dtype" is required (the shape is plan.shape). Only boxes inside the rank's
        block are requested.
    model, plan, shard, batch_size, brightness_clip, normalization_percentiles, group, n_streams
        As in predict_shard.
    affinity_mode : bool, optional
        Three affinity channels (default) or one foreground channel.
    write_block : Callable[[Tuple[int], Tuple[int], numpy.ndarray], None], optional
        Receives every finished box of the rank's output region exactly once as
        write_block(lo, hi, block), block float32 (C, *hi - lo) (or (*hi - lo) if not
        affinity_mode), valid during the call only; z-slabs arrive in z order except the rank's
        first (parked) planes, which arrive last. Without it the rank's whole output region is
        returned as one array.
    keep_input_resident : bool, optional
        Keep the rank's input block in HBM between the passes instead of reading it twice
        (default: if it takes less than a quarter of the free device memory).
    out_dtype : numpy.dtype, optional
        numpy.float32 or numpy.float16 (rounded on the device), as in predict.
    timings : dict, optional
        Filled with "histogram_s", "bands_s" (wall seconds in the exchanges) and "total_s".
    ops : DeviceShardOps, optional
        The device steps (tests substitute numpy restatements).

    Returns
    -------
    numpy.ndarray or None
        (C, *own_dims) (or own_dims) of [shard.own_lo, shard.own_hi), or None with write_block.
    """
    import itertools
    import time

    from aind_exaspim_neuron_segmentation_amd import inference

    t_start = time.perf_counter()
    out_dtype = inference._checked_out_dtype(out_dtype)
    n_channels = 3 if affinity_mode else 1
    if ops is None:
        ops = DeviceShardOps(model, plan, n_channels, batch_size, brightness_clip,
                             inference.DEFAULT_STREAMS if n_streams is None else n_streams,
                             out_dtype == np.float16, copy_threads)
    gshape = plan.shape
    if callable(source) and not hasattr(source, "shape"):
        if dtype is None:
            raise ValueError("a read_box function needs dtype=")
        read_box, src_dtype = source, np.dtype(dtype)
    else:
        src = _BoxSource(source if hasattr(source, "shape") else np.asarray(source))
        if src.shape != tuple(gshape):
            raise ValueError(f"source has shape {src.shape}, the plan is for {tuple(gshape)}")
        read_box, src_dtype = src, src.dtype
    vdtype, convert = ops.storage(src_dtype)
    multi = group is not None and shard.grid[0] * shard.grid[1] > 1
    D, H, W = gshape
    pz, py, px = plan.patch_shape
    trim = plan.trim
    stride = pz - plan.overlap[0]
    z_starts = shard.axis_starts[0][shard.ranges[0][0]:shard.ranges[0][1]]
    yx_starts = list(itertools.product(shard.axis_starts[1][shard.ranges[1][0]:shard.ranges[1][1]],
                                       shard.axis_starts[2][shard.ranges[2][0]:shard.ranges[2][1]]))
    in_lo = shard.input_origin
    in_hi = tuple(o + d for o, d in zip(in_lo, shard.input_dims))
    core_lo = shard.core_origin
    core_hi = tuple(o + d for o, d in zip(core_lo, shard.core_dims))
    acc_lo = shard.accum_origin
    acc_hi = tuple(o + d for o, d in zip(acc_lo, shard.accum_dims))
    AH, AW = shard.accum_dims[1], shard.accum_dims[2]
    own_lo, own_hi = shard.own_lo, shard.own_hi
    own_dims = tuple(h - l for l, h in zip(own_lo, own_hi))

    def read(lo, hi):
        block = read_box(tuple(lo), tuple(hi))
        if tuple(block.shape) != tuple(h - l for l, h in zip(lo, hi)):
            raise ValueError(f"read_box({tuple(lo)}, {tuple(hi)}) returned shape {tuple(block.shape)}")
        return ops.upload(block, convert)

    def timed(key, fn):
        if timings is None:
            return fn()
        ops.synchronize()
        t0 = time.perf_counter()
        out = fn()
        ops.synchronize()
        timings[key] = timings.get(key, 0.0) + time.perf_counter() - t0
        return out

    # ---- pass 1: histogram of the rank's disjoint sub-volume, summed over the ranks -------------
    in_plane_bytes = shard.input_dims[1] * shard.input_dims[2] * np.dtype(vdtype).itemsize
    if keep_input_resident is None:
        keep_input_resident = False
        if ops.device.type == "cuda":
            free, _ = torch.cuda.mem_get_info(ops.device)
            keep_input_resident = shard.input_dims[0] * in_plane_bytes <= free // 4
    chunk = max(1, min(shard.input_dims[0], max(stride, (256 << 20) // max(in_plane_bytes, 1))))
    resident = ops.empty_voxels(shard.input_dims, vdtype) if keep_input_resident else None
    clip, value_dtype = ops.effective_clip(src_dtype, vdtype)
    loaded = [False]

    def core_chunks():
        """The rank's sub-volume in z-chunks (device tensors); fills "resident" on the first walk."""
        z_lo, z_hi = (in_lo[0], in_hi[0]) if resident is not None else (core_lo[0], core_hi[0])
        for z0 in range(z_lo, z_hi, chunk):
            z1 = min(z0 + chunk, z_hi)
            if resident is None:
                yield read((z0, core_lo[1], core_lo[2]), (z1, core_hi[1], core_hi[2]))
                continue
            if not loaded[0]:
                resident[z0 - in_lo[0]:z1 - in_lo[0]].copy_(read((z0, in_lo[1], in_lo[2]), (z1, in_hi[1], in_hi[2])))
            a, b = max(z0, core_lo[0]), min(z1, core_hi[0])
            if b > a:
                yield resident[(slice(a - in_lo[0], b - in_lo[0]),
                                slice(core_lo[1] - in_lo[1], core_hi[1] - in_lo[1]),
                                slice(core_lo[2] - in_lo[2], core_hi[2] - in_lo[2]))]
        loaded[0] = True

    def histogram(pass_index=0, prefix=0):
        hist = torch.zeros(65536, dtype=torch.int64, device=ops.device)
        for part in core_chunks():
            ops.histogram_into(hist, part, vdtype, clip, pass_index, prefix)
        if multi:
            timed("histogram_s", lambda: all_reduce_sum(hist, group))
        return hist.cpu().numpy()

    mn, mx = ops.percentiles(histogram, vdtype, normalization_percentiles, value_dtype, clip)

    # ---- pass 2: the rank's patch layers ----------------------------------------------------------
    band = pz - 2 * trim - stride            # partial sums a layer hands to the next one
    slab_d = min(pz, D)
    # slabs are cut by a rule every rank of a grid row shares (the y exchange pairs them up)
    max_out = max(1, min(D, max(stride + trim, pz), ops.slab_bytes_cap() // (n_channels * H * W * 4)))
    prv_z, nxt_z = (shard.neighbour(-1, 0), shard.neighbour(1, 0)) if multi else (None, None)
    prv_y, nxt_y = (shard.neighbour(0, -1), shard.neighbour(0, 1)) if multi else (None, None)
    y_send = shard.band_box(1) if nxt_y is not None else None
    y_recv = Shard(plan, shard.grid, prv_y).band_box(1) if prv_y is not None else None
    z_recv = Shard(plan, shard.grid, prv_z).band_box(0) if prv_z is not None else None
    # planes [own_lo_z, held_hi) wait for the -z neighbour's band
    held_hi = min(z_recv[1][0], own_hi[0]) if z_recv is not None else own_lo[0]
    held = ops.zeros((n_channels, held_hi - own_lo[0], AH, AW)) if held_hi > own_lo[0] else None

    result = res4 = None
    if write_block is None:
        result = np.empty(((n_channels,) if affinity_mode else ()) + own_dims, dtype=out_dtype)
        res4 = result.reshape((n_channels,) + own_dims)
    drain = ops.make_drain(n_channels * max_out * own_dims[1] * own_dims[2],
                           1 if write_block is not None else max(1, int(copy_threads)))
    acc_flat = [ops.empty((n_channels * slab_d * AH * AW,)) for _ in range(2)]
    in_slab = None
    if resident is None:
        in_slab = [ops.empty_voxels((slab_d,) + tuple(shard.input_dims[1:]), vdtype) for _ in range(2)]
    pbar = None
    if verbose and inference.tqdm is not None:
        pbar = inference.tqdm(total=len(z_starts) * len(yx_starts), desc=f"Predict (rank {shard.rank})")

    def acc_view(k):
        zs = z_starts[k]
        depth = min(zs + pz, D) - zs
        return acc_flat[k % 2][: n_channels * depth * AH * AW].view(n_channels, depth, AH, AW)

    def exchange_y(sums, z0, z1):
        """Trades the y-overlap rows of partial-sum planes [z0, z1) (all accumulator rows) with
        the +-y neighbours and adds what arrives: exchange_output_bands' y phase, slab by slab."""
        ops_list, recv_buf, rows = [], None, None
        if nxt_y is not None:
            lo, hi = y_send
            buf = sums[:, :, lo[1] - acc_lo[1]:hi[1] - acc_lo[1], lo[2] - acc_lo[2]:hi[2] - acc_lo[2]].contiguous()
            if buf.numel():
                ops_list.append(("send", buf, nxt_y))
        if prv_y is not None:
            lo, hi = y_recv
            rows = (slice(None), slice(None), slice(lo[1] - acc_lo[1], hi[1] - acc_lo[1]),
                    slice(lo[2] - acc_lo[2], hi[2] - acc_lo[2]))
            recv_buf = torch.empty((n_channels, z1 - z0, hi[1] - lo[1], hi[2] - lo[2]), dtype=sums.dtype,
                                   device=sums.device)
            if recv_buf.numel():
                ops_list.append(("recv", recv_buf, prv_y))
        if ops_list:
            timed("bands_s", lambda: _p2p(ops_list, group))
        if recv_buf is not None and recv_buf.numel():
            sums[rows] += recv_buf

    own_rows = (slice(None), slice(None), slice(own_lo[1] - acc_lo[1], own_hi[1] - acc_lo[1]),
                slice(own_lo[2] - acc_lo[2], own_hi[2] - acc_lo[2]))

    def consumers(z0, z1):
        lo, hi = (z0, own_lo[1], own_lo[2]), (z1, own_hi[1], own_hi[2])

        def make(view):
            if write_block is not None:
                return [lambda: write_block(lo, hi, view if affinity_mode else view[0])]
            pieces = max(1, min(int(copy_threads), z1 - z0))
            jobs = []
            for i in range(pieces):
                a = z0 + (z1 - z0) * i // pieces
                b = z0 + (z1 - z0) * (i + 1) // pieces
                jobs.append(lambda a=a, b=b: np.copyto(res4[:, a - own_lo[0]:b - own_lo[0]],
                                                       view[:, a - z0:b - z0]))
            return jobs
        return make

    def emit(sums, z0, z1):
        """Owned rows of partial-sum planes [z0, z1), after their exchanges: divide and hand over."""
        for a in range(z0, z1, max_out):
            b = min(a + max_out, z1)

            def fill(out, a=a, b=b):
                out.copy_(sums[:, a - z0:b - z0][own_rows])
                ops.finalize(out, (a, own_lo[1], own_lo[2]))

            drain.emit((n_channels, b - a, own_dims[1], own_dims[2]), fill, consumers(a, b))

    try:
        final_lo = own_lo[0]
        cur = None
        for k, zs in enumerate(z_starts):
            hi = min(zs + pz, D)
            cur = acc_view(k)
            # -- input planes [zs, hi) of the rank's block
            if resident is not None:
                voxels, vox_origin = resident, in_lo
            else:
                slab, old = in_slab[k % 2], in_slab[(k + 1) % 2]
                have = 0
                if k > 0:    # planes shared with the previous layer move on the device
                    zp = z_starts[k - 1]
                    have = max(0, min(zp + pz, D) - zs)
                    if have > 0:
                        slab[:have].copy_(old[zs - zp: zs - zp + have])
                if hi - zs > have:
                    slab[have:hi - zs].copy_(read((zs + have, in_lo[1], in_lo[2]), (hi, in_hi[1], in_hi[2])))
                voxels, vox_origin = slab[: hi - zs], (zs, in_lo[1], in_lo[2])
            # -- this layer's accumulator continues the previous layer's overlap band
            cur.zero_()
            if k > 0 and band > 0:
                zp = z_starts[k - 1]
                b0 = zs + trim
                b1 = min(b0 + band, D, zp + pz)
                if b1 > b0:
                    cur[:, b0 - zs:b1 - zs].copy_(acc_view(k - 1)[:, b0 - zp:b1 - zp])
            ops.run_layer(voxels, vox_origin, src_dtype, vdtype, [(zs, y, x) for y, x in yx_starts], cur,
                          (zs, acc_lo[1], acc_lo[2]), mn, mx, pbar=pbar)
            # -- planes of the rank's region no later layer of it touches
            last = k + 1 == len(z_starts)
            final_hi = own_hi[0] if last else min(z_starts[k + 1] + trim, own_hi[0])
            if final_hi > final_lo:
                sums = ops.zeros((n_channels, final_hi - final_lo, AH, AW))   # planes no patch covers stay 0
                a, b = max(final_lo, zs), min(final_hi, hi)
                if b > a:
                    sums[:, a - final_lo:b - final_lo].copy_(cur[:, a - zs:b - zs])
                # the part below held_hi waits for the -z neighbour's band
                park_hi = min(final_hi, held_hi)
                if held is not None and park_hi > final_lo:
                    held[:, final_lo - own_lo[0]:park_hi - own_lo[0]].copy_(sums[:, :park_hi - final_lo])
                go_lo = max(final_lo, held_hi)
                if final_hi > go_lo:
                    part = sums[:, go_lo - final_lo:]
                    if multi:
                        exchange_y(part, go_lo, final_hi)
                    emit(part, go_lo, final_hi)
                final_lo = final_hi
        # ---- the z-overlap band: mine to +z, the -z neighbour's onto the parked planes --------
        if multi and (nxt_z is not None or prv_z is not None):
            p2p, recv_buf = [], None
            if nxt_z is not None:
                lo, hi = shard.band_box(0)
                send = ops.zeros((n_channels,) + tuple(b - a for a, b in zip(lo, hi)))
                zs = z_starts[-1]
                a, b = max(lo[0], zs), min(hi[0], min(zs + pz, D))
                if b > a:      # (planes the last layer produced; the rest of the box is its trimmed margin)
                    send[:, a - lo[0]:b - lo[0]].copy_(cur[:, a - zs:b - zs])
                if send.numel():
                    p2p.append(("send", send, nxt_z))
            if prv_z is not None:
                lo, hi = z_recv
                recv_buf = torch.empty((n_channels,) + tuple(b - a for a, b in zip(lo, hi)),
                                       dtype=torch.float32, device=ops.device)
                if recv_buf.numel():
                    p2p.append(("recv", recv_buf, prv_z))
            if p2p:
                timed("bands_s", lambda: _p2p(p2p, group))
            if held is not None:
                if recv_buf is not None and recv_buf.numel():
                    lo, hi = z_recv
                    zb = min(hi[0], own_hi[0])          # (the box never reaches past the rank's region: Shard checks)
                    held[:, lo[0] - own_lo[0]:zb - own_lo[0],
                         lo[1] - acc_lo[1]:hi[1] - acc_lo[1],
                         lo[2] - acc_lo[2]:hi[2] - acc_lo[2]] += recv_buf[:, :zb - lo[0]]
                exchange_y(held, own_lo[0], held_hi)
                emit(held, own_lo[0], held_hi)
        elif held is not None:      # (not reached: held planes exist only with a -z neighbour)
            emit(held, own_lo[0], held_hi)
        drain.drain()
    finally:
        drain.close()
        if pbar is not None:
            pbar.close()
    if timings is not None:
        timings["total_s"] = time.perf_counter() - t_start
    return result
