set -e
for v in 0 10 3 5; do
  echo "== variant $v inc.3"; timeout -k 10 120 tools/conv_trace 32 0 32 96 16 gpurun_out/trace_inc3_v$v.bin $v
  echo "== variant $v up4.0"; timeout -k 10 120 tools/conv_trace 32 32 32 96 16 /dev/null $v
done
echo "== up3.3 (64->32 @48)"; timeout -k 10 120 tools/conv_trace 64 0 32 48 16 /dev/null 0
echo "== up3.3 (64->32 @48) one tile per wg"; timeout -k 10 120 tools/conv_trace 64 0 32 48 16 /dev/null 10
