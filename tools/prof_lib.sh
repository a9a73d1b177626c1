# rocprofv3 kernel stats of a few steps with a variant library (run on the MI355X box):
#   tools/prof_lib.sh "base thin8" <grep pattern> [bench args]
# "base" is the in-tree library; other names are csrc/build/variants/lib_<name>.so (see tools/ab_lib.sh).
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/aind_exaspim_neuron_segmentation_amd/csrc
NAMES=$1; PAT=$2; shift; shift
for v in $NAMES; do
  if [ "$v" = base ]; then unset EXASPIM_LIB; else export EXASPIM_LIB=$L/build/variants/lib_$v.so; fi
  echo "== $v"
  rm -rf gpurun_out/prof_lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lib -- python3 bench.py --size 512 --steps 2 --warmup 1 --no-cpu-baseline --no-host-to-host --no-parity --no-configs --pipelined-streams 0 "$@" > gpurun_out/prof_lib.log 2>&1 || { tail -5 gpurun_out/prof_lib.log; exit 1; }
  python profiles/summarize.py stats gpurun_out/prof_lib/*/*_kernel_stats.csv gpurun_out/prof_lib_$v.txt > /dev/null
  rm -rf gpurun_out/prof_lib
  grep -E "$PAT" gpurun_out/prof_lib_$v.txt | cut -c1-75,100-150
done
