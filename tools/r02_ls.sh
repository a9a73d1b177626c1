cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CONV_TRACE_REPEAT=400
for st in 1 2 3 1 2 3; do
  echo "== stride $st: inc.3 / up4.0 shapes"; tools/conv_trace_ab$st 32 0 32 96 16 /dev/null 0 | head -1; tools/conv_trace_ab$st 32 32 32 96 16 /dev/null 0 | head -1
done
