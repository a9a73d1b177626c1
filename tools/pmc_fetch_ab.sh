# HBM read bytes per launch of the z-column kernels for two library builds (MI355X box):
#   tools/pmc_fetch_ab.sh "base new"      ("base" = the in-tree library, others build/variants/lib_<name>.so)
# The variant is selected with EXASPIM_LIB; the product library is never overwritten.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/aind_exaspim_neuron_segmentation_amd/csrc
for v in $1; do
  if [ "$v" = base ]; then unset EXASPIM_LIB; else export EXASPIM_LIB=$L/build/variants/lib_$v.so; fi
  rm -rf gpurun_out/pmc_f
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --size 256 --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host --no-parity --pipelined-streams 0 > /dev/null 2>&1
  echo "== $v"
  python - <<'PY'
import csv, glob, collections
d = collections.defaultdict(list)
for path in glob.glob("gpurun_out/pmc_f/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == "FETCH_SIZE" and "zpipe" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print(f"{k:42s} n={len(v):3d}  read {2 * sum(v) / len(v) * 1024 / 1e6:8.1f} MB per launch")
PY
  rm -rf gpurun_out/pmc_f
done
