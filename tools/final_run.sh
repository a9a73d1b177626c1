# Round-end measurement set (run on the MI355X box): PMC traffic, kernel stats, bench lines.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_final
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --size 256 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --size 256 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_prof_final.log 2>&1
python profiles/summarize.py stats gpurun_out/prof_final/*/*_kernel_stats.csv gpurun_out/final_kernel_stats.txt > /dev/null
python profiles/summarize.py pmc gpurun_out/pmc_fetch/*/*_counter_collection.csv gpurun_out/pmc_write/*/*_counter_collection.csv gpurun_out/final_pmc_hbm_bf16.json bf16 > /dev/null
cp gpurun_out/final_pmc_hbm_bf16.json profiles/r01_pmc_hbm_bf16.json
timeout -k 10 600 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-200
timeout -k 10 300 python bench.py --dtype fp16 --no-cpu-baseline > gpurun_out/bench_fp16.log 2>&1; tail -1 gpurun_out/bench_fp16.log | cut -c1-200
timeout -k 10 300 python bench.py --dtype fp32 --size 512 --batch 8 --no-cpu-baseline > gpurun_out/bench_fp32.log 2>&1; tail -1 gpurun_out/bench_fp32.log | cut -c1-200
