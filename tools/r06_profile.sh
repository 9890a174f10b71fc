# Round-6 measurement pass on one MI355X box (through gpurun from the repository root):   bash tools/r06_profile.sh [TAG]   -> gpurun_out/TAG_*
# The driver-flag bench line, rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE (separate passes) and the SQ issue / stall sets on
# chain_v4_kernel, the other configs' lines and kernel stats, the kpn graph sweep, the drop-in per-call table, the moved-bytes PMC passes.
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_flags.json 2> $O/${TAG}_bench_driver_flags.err
python3 tools/bench_configs.py c3 c3big c4 c5 fir fft firshapes c4gen u8chain u8c4 srcsmall > $O/${TAG}_other_configs_bench_lines.txt 2>&1
python3 tools/dropin_latency.py > $O/${TAG}_dropin_calls.txt 2>&1
timeout 900 tests/_build/kpn_tests bench_c2_sweep 0.3 3 13 28 > $O/${TAG}_kpn_sweep.txt 2>&1
for src in resident synth; do for snk in checksum drop; do for k in "24 2000" "28 200"; do
  timeout 120 tests/_build/kpn_tests bench_c2 $k 4 $src $snk 0 0 >> $O/${TAG}_kpn_variants.txt 2>&1
done; done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_p_stats -- python3 $R/bench.py --no-cpu-baseline --no-graph-leg > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_p_stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_p_other -- python3 $R/tools/bench_configs.py c3 c3big c4 c5 fir fft c4gen u8c4 > $O/${TAG}_other_under_rocprof.txt 2> $O/${TAG}_p_other.err
B="python3 $R/bench.py --steps 20 --warmup 5 --steady 0 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_p_fetch -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_p_write -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/${TAG}_p_sq1 -- $B > /dev/null 2> $O/${TAG}_p_sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${TAG}_p_sq2 -- $B > /dev/null 2> $O/${TAG}_p_sq2.err
cd $R
{
  echo "# chain_v4_kernel (fmaf build), mean per launch (bench.py --steps 20 --warmup 5 --steady 0, rocprofv3 --pmc, one pass per line group)"
  python3 profiles/pmc_summary.py "chain_v4_kernel<127, 5, true, 2, 8, false, true, false, false>" $O/${TAG}_p_fetch $O/${TAG}_p_write $O/${TAG}_p_sq1 $O/${TAG}_p_sq2
} > $O/${TAG}_chain_pmc_summary.txt 2>&1
cat $O/${TAG}_chain_pmc_summary.txt
find $O/${TAG}_p_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_chain_v4_kernel_stats.csv
find $O/${TAG}_p_other -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_other_configs_kernel_stats.csv
head -6 $O/${TAG}_chain_v4_kernel_stats.csv | cut -c1-200
bash tools/pmc_moved.sh 06 > $O/${TAG}_pmc_moved.txt 2>&1
cut -c1-300 $O/${TAG}_bench_driver_flags.json
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
