# Package power and shader clock sampled every half second beside real steps (run on the MI355X box):
#   bash tools/power_steps.sh [streams]     -> gpurun_out/power_steps_<streams>.txt
cd $GRAFT_REPO_ROOT
S=${1:-1}
OUT=gpurun_out/power_steps_$S.txt
rocm-smi --showmaxpower 2>&1 | grep -E "Max" > $OUT
python tools/layer_times.py --size 1024 --steps 12 --streams $S --mask 0 > gpurun_out/power_steps_run.log 2>&1 &
BG=$!
while kill -0 $BG 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "GPU\[0\].*(Power|sclk)" | sed -e 's/.*sclk clock level: [^(]*(\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/power \1/' | tr '\n' ' ' >> $OUT; echo >> $OUT
  sleep 0.5
done
wait $BG
grep ms_per_step gpurun_out/power_steps_run.log >> $OUT
python - $OUT <<'PY'
import re, sys
rows = [(int(m.group(1)), float(m.group(2))) for m in (re.search(r"sclk (\d+) power ([\d.]+)", l) for l in open(sys.argv[1])) if m]
busy = [r for r in rows if r[1] > 600]
print(f"{len(rows)} samples, {len(busy)} above 600 W")
if busy:
    p = sorted(r[1] for r in busy); c = sorted(r[0] for r in busy)
    print(f"power W: min {p[0]:.0f} median {p[len(p)//2]:.0f} max {p[-1]:.0f}; sclk MHz: min {c[0]} median {c[len(c)//2]} max {c[-1]}")
PY
tail -1 $OUT
