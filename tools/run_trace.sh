set -e
export CONV_TRACE_REPEAT=3000
for v in 0 41 42; do
echo "== down4.0 (256->256 @6) v$v"; timeout -k 10 60 tools/conv_trace 256 0 256 6 16 /dev/null $v
done
