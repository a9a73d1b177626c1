cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CONV_TRACE_REPEAT=200
echo "== down1.3 (64->64 @48)"; tools/conv_trace 64 0 64 48 16 /tmp/t1.bin 0 && python tools/analyze_trace.py /tmp/t1.bin 0 4
echo "== up3.0 (64+64->64 @48)"; tools/conv_trace 64 64 64 48 16 /tmp/t2.bin 0 | head -2
echo "== down2.3 (128->128 @24)"; tools/conv_trace 128 0 128 24 16 /tmp/t3.bin 0 | head -2
