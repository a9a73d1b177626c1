cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in 1 2 3 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-to-host --no-parity --steps 3 --streams $s 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams=$s ms_per_step %.1f value %.4e roofline avg_launch_ms %.4f frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
