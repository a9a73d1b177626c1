# A/B of one environment switch inside bench.py steps (run on the MI355X box):
#   tools/ab_env.sh VAR "v1 v2" [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; shift 2
for v in $VALS $VALS; do
  env $VAR=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-to-host --pipelined-streams 0 --steps 3 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
p=d.get('parity') or {}
print('$VAR=$v ms_per_step %.1f  dominant %.4f ms frac %.3f  parity max %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], p.get('max')))"
done
