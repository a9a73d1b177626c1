"""Summarises a tools/conv_trace dump: per-phase cycle counts of a workgroup's
life and how busy each CU's wave slots are (diagnostic, not part of the library)."""
import sys

import numpy as np


def main(path, ms=None, nchunks=0):
    t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4, 16).astype(np.int64)
    t = t[t[:, 0, 14] > 0]   # records of workgroups / tiles that ran
    nwg = t.shape[0]
    hw = t[:, :, 15]
    xcc = (hw >> 32) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    simd = (hw >> 4) & 3
    cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print(f"{nwg} workgroups, {len(np.unique(cukey[:, 0]))} CUs seen, simds of waves 0..3 (first WG): {simd[0]}")
    ev = t[:, :, :15]
    nch = nchunks or 1 + max(c for c in range(4) if (ev[:, 0, 3 + 3 * c] > 0).all())
    span = ev[:, :, 14].max() - ev[:, :, 0].min()
    print(f"chunks {nch}; stamp span {span} ticks" + (f" = {span / (ms * 1e3):.1f} ticks/us" if ms else ""))

    def stat(name, d):
        d = d.reshape(-1)
        print(f"  {name:34s} median {np.median(d):8.0f}  mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f}")

    stat("life (0 -> end)", ev[:, :, 14] - ev[:, :, 0])
    stat("prologue: issue loads", ev[:, :, 1] - ev[:, :, 0])
    stat("prologue: wait+store+barrier", ev[:, :, 2] - ev[:, :, 1])
    prev = ev[:, :, 2]
    for c in range(nch):
        stat(f"chunk {c}: tap loop", ev[:, :, 3 + 3 * c] - prev)
        stat(f"chunk {c}: barrier 1", ev[:, :, 4 + 3 * c] - ev[:, :, 3 + 3 * c])
        if c + 1 < nch:
            stat(f"chunk {c}: store + barrier 2", ev[:, :, 5 + 3 * c] - ev[:, :, 4 + 3 * c])
            prev = ev[:, :, 5 + 3 * c]
    stat("epilogue", ev[:, :, 14] - ev[:, :, 4 + 3 * (nch - 1)])
    if nch == 2 and (ev[:, 0, 9] > 0).all():   # quarter marks inside the first chunk's tap loop
        marks = [ev[:, :, 2], ev[:, :, 9], ev[:, :, 10], ev[:, :, 11], ev[:, :, 3]]
        for q in range(4):
            stat(f"chunk 0 loop, quarter {q}", marks[q + 1] - marks[q])

    # per-CU occupancy: time with k workgroups resident / in their tap loops
    res_frac = np.zeros(4)
    loop_frac = np.zeros(4)
    gaps = []
    for key in np.unique(cukey[:, 0]):
        sel = np.flatnonzero(cukey[:, 0] == key)
        e = ev[sel]
        t0, t1 = e[:, :, 0].min(), e[:, :, 14].max()
        pts = []
        for w in e:
            pts.append((w[:, 0].min(), 1, 0))
            pts.append((w[:, 14].max(), -1, 0))
            prev = w[0, 2]
            for c in range(nch):
                pts.append((prev, 0, 1))
                pts.append((w[0, 3 + 3 * c], 0, -1))
                if c + 1 < nch:
                    prev = w[0, 5 + 3 * c]
        pts.sort()
        r = l = 0
        last = t0
        for x, dr, dl in pts:
            res_frac[min(r, 3)] += x - last
            loop_frac[min(l, 3)] += x - last
            last = x
            r += dr
            l += dl
        starts = np.sort(e[:, 0, 0])
        ends = np.sort(e[:, :, 14].max(axis=1))
        gaps.append(len(sel))
    print("CU time with k workgroups resident (k=0,1,2,3+):", np.round(res_frac / res_frac.sum(), 3))
    print("CU time with k workgroups (wave 0) inside a tap loop:", np.round(loop_frac / loop_frac.sum(), 3))
    print("workgroups per CU: min %d max %d" % (min(gaps), max(gaps)))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 and float(sys.argv[2]) > 0 else None,
         int(sys.argv[3]) if len(sys.argv) > 3 else 0)
