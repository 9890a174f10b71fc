cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/p_stats.json 2> $R/gpurun_out/p_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_fetch -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_write -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_cfg -- python3 $R/tools/bench_configs.py c3 c4 c5 fir fft > $R/gpurun_out/p_cfg.txt 2>&1
cd $R
python3 profiles/pmc_summary.py chain_v4 gpurun_out/p_fetch gpurun_out/p_write
cat gpurun_out/p_stats.json | cut -c1-400
# HBM traffic of the 65536-point overlap-save passes (Infinity-Cache residency of the 64 MiB work buffers)
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_c5_fetch -- python3 $R/tools/bench_configs.py c5only > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_c5_write -- python3 $R/tools/bench_configs.py c5only > /dev/null 2>&1
cd $R
for k in "fftbig_first_kernel<false>" ovsave64k_mid_wave ovsave64k_last_wave; do echo "== $k"; python3 profiles/pmc_summary.py "$k" gpurun_out/p_c5_fetch gpurun_out/p_c5_write; done
