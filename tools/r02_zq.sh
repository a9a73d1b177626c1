cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CONV_TRACE_REPEAT=400
for z in 0 1; do
  echo "== ZPAIR=$z inc.3 shape"; EXASPIM_ZPAIR=$z timeout -k 10 60 tools/conv_trace 32 0 32 96 16 /dev/null 0 | head -1
  echo "== ZPAIR=$z up4.0 shape"; EXASPIM_ZPAIR=$z timeout -k 10 60 tools/conv_trace 32 32 32 96 16 /dev/null 0 | head -1
done
