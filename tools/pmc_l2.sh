# L2 (TCC) requests / hits / misses per kernel of one 256^3 step (run on the MI355X box):
#   bash tools/pmc_l2.sh [grep pattern]    -> gpurun_out/pmc_l2.txt
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-host-to-host --no-parity --no-configs --pipelined-streams 0"
rm -rf gpurun_out/pmc_l2
timeout -k 10 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_l2 -- python3 bench.py --size 256 --steps 1 --warmup 0 $Q > gpurun_out/pmc_l2.log 2>&1
python - "$@" <<'PY'
import csv, glob, sys, collections, re
pat = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for f in glob.glob('gpurun_out/pmc_l2/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:90]
        if pat and not pat.search(k): continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'TCC_REQ_sum':
            n[k] += 1
            dur[k] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) if 'End_Timestamp' in r and r['End_Timestamp'] else 0
with open('gpurun_out/pmc_l2.txt', 'w') as out:
    for k in sorted(acc, key=lambda k: -acc[k]['TCC_REQ_sum']):
        v = acc[k]; l = max(n[k], 1)
        line = '%-90s launches %3d  req/launch %.3e  hit %.3e  miss %.3e  hit rate %.3f  us/launch %.1f' % (k, n[k], v['TCC_REQ_sum'] / l, v['TCC_HIT_sum'] / l, v['TCC_MISS_sum'] / l, v['TCC_HIT_sum'] / max(v['TCC_HIT_sum'] + v['TCC_MISS_sum'], 1), dur[k] / l / 1e3)
        print(line); out.write(line + '\n')
PY
rm -rf gpurun_out/pmc_l2
