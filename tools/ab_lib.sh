# A/B of library variants inside real steps (run on the MI355X box):
#   tools/ab_lib.sh "base ls12 ls24" [layer_times args]
# "base" is the in-tree library; other names are csrc/build/variants/lib_<name>.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=aind_exaspim_neuron_segmentation_amd/csrc
cp $L/libexaspim_affinity.so $L/build/variants/lib_base.so
NAMES=$1; shift
for v in $NAMES $NAMES; do
  cp $L/build/variants/lib_$v.so $L/libexaspim_affinity.so
  echo "== $v"; timeout -k 10 200 python tools/layer_times.py "$@" 2>&1 | tail -2
done
cp $L/build/variants/lib_base.so $L/libexaspim_affinity.so
