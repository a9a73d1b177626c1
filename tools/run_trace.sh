set -e
export CONV_TRACE_REPEAT=3000
for b in conv_trace_ab0 conv_trace conv_trace_ab0 conv_trace; do
echo "== $b inc.3"; timeout -k 10 120 tools/$b 32 0 32 96 16 /dev/null 0
echo "== $b up4.0"; timeout -k 10 120 tools/$b 32 32 32 96 16 /dev/null 0
done
