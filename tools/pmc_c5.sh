# SQ counters for the 65536-point kernels (run on the GPU box: bash tools/pmc_c5.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/c5_sq1 -- python3 $R/tools/bench_configs.py c5 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/c5_sq2 -- python3 $R/tools/bench_configs.py c5 > /dev/null 2>&1
cd $R
for k in "fftbig_first_kernel<false>" ovsave64k_mid_wave ovsave64k_last_wave; do echo "== $k"; python3 profiles/pmc_summary.py "$k" gpurun_out/c5_sq1 gpurun_out/c5_sq2; done
