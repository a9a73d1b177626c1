# Samples package power and shader clock beside a sustained run (diagnostic).
#   bash tools/power_probe.sh <command ...>      e.g.
#   CONV_TRACE_REPEAT=6000 bash tools/power_probe.sh tools/conv_trace 32 32 32 96 16 /dev/null 0
#   bash tools/power_probe.sh tools/mfma_power 11 7
# Ablation builds of the convolution: add -DEXASPIM_ABLATE=1|2|4 (no prefetch loads / no
# output stores / no LDS staging writes) to the conv_trace compile line in conv_trace.hip.
rocm-smi --showmaxpower 2>&1 | grep -E "Max"
"$@" &
BG=$!
sleep 5
for i in 1 2 3; do rocm-smi --showpower --showclocks 2>&1 | grep -E "GPU\[0\].*(Power|sclk)" | tr '\n' ' '; echo; sleep 1; done
wait $BG
