# Round 2, first GPU pass: all GPU tests, the default bench line, a 4-rank gloo rehearsal of the
# self-launching bench on the one GPU, and a kernel-stats profile of the fp16 default.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02_gpu_tests_1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02_gpu_tests_1.log
timeout -k 10 500 python bench.py > gpurun_out/r02_bench_1.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r02_bench_1.log | cut -c1-1500
EXASPIM_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 4 --size 256 --steps 1 --warmup 1 > gpurun_out/r02_bench_gloo4.log 2>&1; echo "gloo4 rc=$?"; tail -1 gpurun_out/r02_bench_gloo4.log | cut -c1-600
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof1 -- python3 bench.py --no-cpu-baseline --no-host-to-host --no-parity > gpurun_out/r02_bench_prof1.log 2>&1; echo "prof rc=$?"
python profiles/summarize.py stats gpurun_out/r02_prof1/*/*_kernel_stats.csv gpurun_out/r02_kernel_stats_1.txt > /dev/null; head -30 gpurun_out/r02_kernel_stats_1.txt
