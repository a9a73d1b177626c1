set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "unet or predict_fp32 or conv_transpose or half_width" > gpurun_out/gpu_tests_10.log 2>&1; tail -2 gpurun_out/gpu_tests_10.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --size 256 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --size 256 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/bench_r01_final.log 2>&1; tail -1 gpurun_out/bench_r01_final.log | cut -c1-200
