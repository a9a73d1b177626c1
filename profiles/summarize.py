"""
Summarises rocprofv3 CSV output (kernel stats and PMC passes) into the small
text/JSON files committed under profiles/.

  python profiles/summarize.py stats <kernel_stats.csv> <out.txt>
  python profiles/summarize.py pmc <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [dtype]
  python profiles/summarize.py util <mfma_counter_collection.csv> <lds_counter_collection.csv> <out.json> [dtype]

HBM traffic per launch follows MI355X_MICROARCH.md section HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
coalesced reads, so reads = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.
"""

import collections
import csv
import json
import sys


def short(name):
    return name.replace("exaspim::", "").replace("void ", "")


def stats(path, out):
    rows = list(csv.DictReader(open(path)))
    lines = [f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>7s}"]
    for r in rows:
        lines.append(
            f"{short(r['Name'])[:100]:100s} {r['Calls']:>7s} "
            f"{float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['AverageNs']) / 1e3:10.1f} "
            f"{float(r['Percentage']):7.3f}"
        )
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))


def pmc(fetch_path, write_path, out, dtype):
    def agg(path, counter):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                d[short(r["Kernel_Name"]).split("(")[0]].append(float(r["Counter_Value"]))
        return d

    f, w = agg(fetch_path, "FETCH_SIZE"), agg(write_path, "WRITE_SIZE")
    table = {}
    for k in sorted(f):
        fv = sum(f[k]) / len(f[k])
        wv = sum(w[k]) / len(w[k]) if k in w else 0.0
        table[k] = {
            "launches": len(f[k]),
            "fetch_size_kib_avg": fv,
            "write_size_kib_avg": wv,
            "hbm_read_bytes_per_launch": 2.0 * fv * 1024.0,
            "hbm_write_bytes_per_launch": wv * 1024.0,
            "hbm_bytes_per_launch": 2.0 * fv * 1024.0 + wv * 1024.0,
        }
        print(f"{k[:80]:80s} n={len(f[k]):4d} read {2 * fv * 1024 / 1e6:9.1f} MB  write {wv * 1024 / 1e6:9.1f} MB")
    json.dump({"dtype": dtype, "kernels": table}, open(out, "w"), indent=1)


def util(mfma_path, lds_path, out, dtype):
    """Matrix-pipe utilisation and LDS bank-conflict share per kernel symbol from two PMC passes:
    --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE and --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE.
    MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
    def agg(path):
        d = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"]).split("(")[0]
            d[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k].add(r["Dispatch_Id"])
        return d, n

    m, mn = agg(mfma_path)
    l, _ = agg(lds_path)
    table = {}
    for k in sorted(m):
        gui = m[k].get("GRBM_GUI_ACTIVE", 0.0)
        busy = m[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if gui <= 0 or busy <= 0:
            continue
        idx = l.get(k, {}).get("SQ_LDS_IDX_ACTIVE", 0.0)
        table[k] = {
            "launches": len(mn[k]),
            "mfma_busy_pct_of_simd_cycles": round(100.0 * busy / (gui / 8.0 * 1024.0), 1),
            "lds_bank_conflict_pct_of_lds_active_cycles":
                round(100.0 * l[k].get("SQ_LDS_BANK_CONFLICT", 0.0) / idx, 1) if idx > 0 else None,
        }
        print(f"{k[:80]:80s} n={len(mn[k]):4d} mfma busy {table[k]['mfma_busy_pct_of_simd_cycles']:5.1f} %  "
              f"lds conflicts {table[k]['lds_bank_conflict_pct_of_lds_active_cycles']} %")
    note = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE and --pmc SQ_LDS_BANK_CONFLICT "
            "SQ_LDS_IDX_ACTIVE (separate passes, --kernel-trace only) over bench.py --size 256 --steps 1 "
            "--warmup 0; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)")
    json.dump({"dtype": dtype, "note": note, "kernels": table}, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "util":
        util(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "fp16")
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "bf16")
