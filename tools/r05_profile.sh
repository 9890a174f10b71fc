# Round-5 measurement pass on the MI355X box (run through gpurun from the repository root):
#   bash tools/r05_profile.sh [TAG] [notests]        -> gpurun_out/TAG_*
# GPU tests, the driver-flag bench line, the default bench line, then rocprofv3 passes: kernel stats of bench.py and of the other
# configs, FETCH_SIZE / WRITE_SIZE (separate passes) and the SQ issue/stall sets on chain_v4_kernel, SQ sets on the two C3 kernels.
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$2" != "notests" ]; then
  timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.txt
  tail -3 $O/${TAG}_pytest.txt
fi
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_flags.json 2> $O/${TAG}_bench_driver_flags.err
python3 bench.py --no-cpu-baseline > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
python3 tools/bench_configs.py c3 c4 c5 fir fft firshapes srcgen srcsmall ingest u8chain > $O/${TAG}_other_configs_bench_lines.txt 2>&1
python3 tools/bench_bigfft.py > $O/${TAG}_bigfft_bench_lines.txt 2>&1
python3 tools/bench_configs.py c3big > $O/${TAG}_c3big_lines.txt 2>&1
python3 tools/bench_configs.py hipfft > $O/${TAG}_vendor_fft_yardstick.txt 2>&1   # same-hardware yardstick, never the engine
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_p_stats -- python3 $R/bench.py --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_p_stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_p_other -- python3 $R/tools/bench_configs.py c3 c4 c5 fir fft > $O/${TAG}_other_under_rocprof.txt 2> $O/${TAG}_p_other.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_p_fetch -- python3 $R/bench.py --steps 20 --warmup 5 --steady 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_p_write -- python3 $R/bench.py --steps 20 --warmup 5 --steady 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/${TAG}_p_sq1 -- python3 $R/bench.py --steps 20 --warmup 5 --steady 0 --no-cpu-baseline > /dev/null 2> $O/${TAG}_p_sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${TAG}_p_sq2 -- python3 $R/bench.py --steps 20 --warmup 5 --steady 0 --no-cpu-baseline > /dev/null 2> $O/${TAG}_p_sq2.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_p_sq3 -- python3 $R/bench.py --steps 20 --warmup 5 --steady 0 --no-cpu-baseline > /dev/null 2> $O/${TAG}_p_sq3.err
cd $R
{
  echo "# chain_v4_kernel (fmaf build), mean per launch (bench.py --steps 20 --warmup 5 --steady 0, rocprofv3 --pmc, one pass per line group)"
  python3 profiles/pmc_summary.py "chain_v4_kernel<127, 5, true, 2, 8, false, true, false>" $O/${TAG}_p_fetch $O/${TAG}_p_write $O/${TAG}_p_sq1 $O/${TAG}_p_sq2 $O/${TAG}_p_sq3
} > $O/${TAG}_chain_pmc_summary.txt 2>&1
cat $O/${TAG}_chain_pmc_summary.txt
find $O/${TAG}_p_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_chain_v4_kernel_stats.csv
find $O/${TAG}_p_other -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_other_configs_kernel_stats.csv
head -4 $O/${TAG}_chain_v4_kernel_stats.csv
bash tools/shape_pmc.sh ${TAG}_c3exact src_window_rb src 256 20 > /dev/null 2>&1
bash tools/shape_pmc.sh ${TAG}_c3fast src_window_fastp2 srcfast 256 20 > /dev/null 2>&1
cat $O/${TAG}_c3exact_pmc_summary.txt $O/${TAG}_c3fast_pmc_summary.txt | grep -v "^/opt"
cut -c1-400 $O/${TAG}_bench_driver_flags.json
# keep the bulky traces out of the merge-back
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
