cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CONV_TRACE_REPEAT=300
echo "== inc.3 shape (32->32 @96)"; timeout -k 10 60 tools/conv_trace 32 0 32 96 16 gpurun_out/trace_inc3.bin 0 && python tools/analyze_trace.py gpurun_out/trace_inc3.bin
echo "== up4.0 shape (32+32->32 @96)"; timeout -k 10 60 tools/conv_trace 32 32 32 96 16 gpurun_out/trace_up40.bin 0 && python tools/analyze_trace.py gpurun_out/trace_up40.bin
