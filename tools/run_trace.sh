set -e
export CONV_TRACE_REPEAT=2000
echo "== down1.3 (64->64 @48)"; timeout -k 10 60 tools/conv_trace 64 0 64 48 16 gpurun_out/trace_d13.bin 0
echo "== down2.3 (128->128 @24)"; timeout -k 10 60 tools/conv_trace 128 0 128 24 16 gpurun_out/trace_d23.bin 0
