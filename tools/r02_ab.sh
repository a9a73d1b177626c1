cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for z in 0 1 0 1; do
  EXASPIM_ZPAIR=$z timeout -k 10 300 python bench.py --no-cpu-baseline --no-host-to-host --no-parity --steps 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ZPAIR=$z ms_per_step %.1f  roofline avg_launch_ms %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof3 -- python3 bench.py --no-cpu-baseline --no-host-to-host --no-parity > /dev/null 2>&1
python profiles/summarize.py stats gpurun_out/r02_prof3/*/*_kernel_stats.csv gpurun_out/r02_kernel_stats_3.txt > /dev/null; head -12 gpurun_out/r02_kernel_stats_3.txt; rm -rf gpurun_out/r02_prof3
