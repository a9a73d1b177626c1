cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests_2.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02_gpu_tests_2.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof2 -- python3 bench.py --no-cpu-baseline --no-host-to-host --no-parity > gpurun_out/r02_bench_prof2.log 2>&1; echo "prof rc=$?"; tail -1 gpurun_out/r02_bench_prof2.log | cut -c1-400
python profiles/summarize.py stats gpurun_out/r02_prof2/*/*_kernel_stats.csv gpurun_out/r02_kernel_stats_2.txt > /dev/null; head -14 gpurun_out/r02_kernel_stats_2.txt; rm -rf gpurun_out/r02_prof2
