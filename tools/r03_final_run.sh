# Round-3 measurement set (run on the MI355X box): PMC traffic + utilisation, kernel stats, the bench line,
# per-layer times. Every step runs under its own timeout; the script stops at the first failing step.
#   bash tools/r03_final_run.sh            -> files under gpurun_out/ (copy the r03_* ones to profiles/)
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
Q="--no-cpu-baseline --no-host-to-host --no-parity --no-configs --pipelined-streams 0"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_lds gpurun_out/prof_final
for dt in fp16 bf16; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --dtype $dt --size 256 --steps 1 --warmup 0 $Q > gpurun_out/pmc_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --dtype $dt --size 256 --steps 1 --warmup 0 $Q > gpurun_out/pmc_write.log 2>&1
  python profiles/summarize.py pmc gpurun_out/pmc_fetch/*/*_counter_collection.csv gpurun_out/pmc_write/*/*_counter_collection.csv gpurun_out/r03_pmc_hbm_$dt.json $dt > gpurun_out/r03_pmc_hbm_$dt.txt
  rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
  echo "pmc hbm $dt done"
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --size 256 --steps 1 --warmup 0 $Q > gpurun_out/pmc_mfma.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --size 256 --steps 1 --warmup 0 $Q > gpurun_out/pmc_lds.log 2>&1
python profiles/summarize.py util gpurun_out/pmc_mfma/*/*_counter_collection.csv gpurun_out/pmc_lds/*/*_counter_collection.csv gpurun_out/r03_pmc_util_fp16.json fp16 > gpurun_out/pmc_util.log 2>&1
rm -rf gpurun_out/pmc_mfma gpurun_out/pmc_lds
echo "pmc util done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py $Q > gpurun_out/bench_prof_final.log 2>&1
python profiles/summarize.py stats gpurun_out/prof_final/*/*_kernel_stats.csv gpurun_out/r03_bench_fp16_1024_kernel_stats.txt > /dev/null
rm -rf gpurun_out/prof_final
echo "kernel stats done"
python tools/layer_times.py --size 1024 > gpurun_out/r03_layer_times.txt 2>&1; cat gpurun_out/r03_layer_times.txt
cp gpurun_out/r03_pmc_hbm_fp16.json gpurun_out/r03_pmc_hbm_bf16.json profiles/     # the bench line quotes them as roofline.traffic
timeout -k 10 900 python bench.py > gpurun_out/r03_bench_default.log 2>&1; tail -1 gpurun_out/r03_bench_default.log > gpurun_out/r03_bench_default.json; cut -c1-400 gpurun_out/r03_bench_default.json
