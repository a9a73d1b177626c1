# samples clocks and power while a kernel runs (diagnostic)
for ab in 0 1 2 4 7; do
  echo "== ablation $ab (1 = no prefetch loads, 2 = no stores, 4 = no LDS staging)"
  CONV_TRACE_REPEAT=6000 tools/conv_trace_ab$ab 32 32 32 96 16 /dev/null 0 &
  BG=$!
  sleep 6
  rocm-smi --showpower --showclocks 2>&1 | grep -E "GPU\[0\].*(Power|sclk)" | tr '\n' ' '; echo
  wait $BG
done
