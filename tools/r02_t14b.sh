cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CONV_TRACE_REPEAT=300
for sh in "32 0 64 48" "64 0 64 48" "64 64 64 48" "64 0 128 24" "128 0 128 24" "128 128 128 24" "128 0 64 24" "128 0 256 12" "256 0 256 12" "256 256 256 12" "256 0 128 12" "256 0 256 6"; do
  echo -n "$sh: "; timeout -k 10 60 tools/conv_trace $sh 16 /dev/null 0 | head -1
done
