# Sustained time and rate of every MFMA convolution of the network (bf16, batch 16, full patches,
# 2000 back-to-back launches each): tools/conv_trace must have been built (see conv_trace.hip).
export CONV_TRACE_REPEAT=2000
run() { printf "%-10s %4s+%-4s -> %-4s @%-3s " "$1" "$2" "$3" "$4" "$5"; tools/conv_trace $2 $3 $4 $5 16 /dev/null 0 | grep untraced | sed 's/untraced: //'; }
run inc.3    32   0  32 96
run down1.0  32   0  64 48
run down1.3  64   0  64 48
run down2.0  64   0 128 24
run down2.3 128   0 128 24
run down3.0 128   0 256 12
run down3.3 256   0 256 12
run down4.0 256   0 256  6
run down4.3 256   0 256  6
run up1.0   256 256 256 12
run up1.3   256   0 128 12
run up2.0   128 128 128 24
run up2.3   128   0  64 24
run up3.0    64  64  64 48
run up3.3    64   0  32 48
run up4.0    32  32  32 96
run up4.3    32   0  32 96
