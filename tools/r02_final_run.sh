# Round-2 measurement set (run on the MI355X box): PMC traffic, kernel stats, bench lines.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_final
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --size 256 --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host --no-parity > gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --size 256 --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host --no-parity > gpurun_out/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --no-cpu-baseline --no-host-to-host --no-parity > gpurun_out/bench_prof_final.log 2>&1
python profiles/summarize.py stats gpurun_out/prof_final/*/*_kernel_stats.csv profiles/r02_bench_fp16_1024_kernel_stats.txt > /dev/null
python profiles/summarize.py pmc gpurun_out/pmc_fetch/*/*_counter_collection.csv gpurun_out/pmc_write/*/*_counter_collection.csv profiles/r02_pmc_hbm_fp16.json fp16 > /dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_final
cp profiles/r02_bench_fp16_1024_kernel_stats.txt profiles/r02_pmc_hbm_fp16.json gpurun_out/
timeout -k 10 600 python bench.py > gpurun_out/r02_bench_default.log 2>&1; tail -1 gpurun_out/r02_bench_default.log > gpurun_out/r02_bench_default.json; cut -c1-300 gpurun_out/r02_bench_default.json
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --no-host-to-host > gpurun_out/r02_bench_bf16.log 2>&1; tail -1 gpurun_out/r02_bench_bf16.log > gpurun_out/r02_bench_bf16_1024.json; cut -c1-200 gpurun_out/r02_bench_bf16_1024.json
timeout -k 10 300 python bench.py --dtype fp32 --size 512 --batch 8 --no-cpu-baseline --no-host-to-host > gpurun_out/r02_bench_fp32.log 2>&1; tail -1 gpurun_out/r02_bench_fp32.log > gpurun_out/r02_bench_fp32_512.json; cut -c1-200 gpurun_out/r02_bench_fp32_512.json
for v in 0 1 2 4 3; do timeout -k 10 60 tools/mfma_shape $v 6; done > gpurun_out/r02_mfma_shape.txt 2>&1; cat gpurun_out/r02_mfma_shape.txt
