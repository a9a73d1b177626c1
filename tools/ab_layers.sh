# A/B of one environment switch with per-layer times inside real steps (MI355X box):
#   tools/ab_layers.sh VAR "v1 v2" [layer_times args]        stops at the first failing run
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; shift 2
for v in $VALS $VALS; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 200 python tools/layer_times.py "$@" > gpurun_out/ab_layers_run.log 2>&1 || { tail -5 gpurun_out/ab_layers_run.log; exit 1; }
  if grep -q "Memory access fault" gpurun_out/ab_layers_run.log; then tail -5 gpurun_out/ab_layers_run.log; exit 1; fi
  tail -2 gpurun_out/ab_layers_run.log
done
