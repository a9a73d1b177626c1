# A/B of library variants inside real steps (run on the MI355X box):
#   tools/ab_lib.sh "base ls12 ls24" [layer_times args]
# "base" is the in-tree library; other names are csrc/build/variants/lib_<name>.so, built with
# `make -C aind_exaspim_neuron_segmentation_amd/csrc variant NAME=<name> VFLAGS=...`. The variant
# is selected with EXASPIM_LIB (aind_exaspim_neuron_segmentation_amd/_native.py): the product
# library is never overwritten. Two alternating repeats; stops at the first failing run.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/aind_exaspim_neuron_segmentation_amd/csrc
NAMES=$1; shift
for v in $NAMES $NAMES; do
  if [ "$v" = base ]; then unset EXASPIM_LIB; else export EXASPIM_LIB=$L/build/variants/lib_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python tools/layer_times.py "$@" > gpurun_out/ab_lib_run.log 2>&1 || { tail -5 gpurun_out/ab_lib_run.log; exit 1; }
  if grep -q "Memory access fault" gpurun_out/ab_lib_run.log; then tail -5 gpurun_out/ab_lib_run.log; exit 1; fi
  tail -2 gpurun_out/ab_lib_run.log
done
