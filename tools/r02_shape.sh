cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 5 2 3 4 6; do bash tools/power_probe.sh tools/mfma_shape $v 8 2>&1 | grep -E "variant|Power|sclk" ; done
