# round 6: SQ issue / stall sets and HBM traffic of the channelizer kernels whose cache policy changed this round (2^28 samples)
R=$GRAFT_REPO_ROOT; cd $R
bash tools/shape_pmc.sh r06_pfbu8 "pfb64_kernel<16, true, true, 1, 3>" pfbu8 64 16 28 6 > /dev/null 2>&1
bash tools/shape_pmc.sh r06_pfb256 "pfb_p2_kernel<8, 16" pfb 256 16 28 6 > /dev/null 2>&1
bash tools/shape_pmc.sh r06_pfb64 "pfb64_kernel<16, true, true, 0, 0>" pfb 64 16 28 6 > /dev/null 2>&1
cat gpurun_out/r06_pfbu8_pmc_summary.txt gpurun_out/r06_pfb256_pmc_summary.txt gpurun_out/r06_pfb64_pmc_summary.txt | grep -v "^/opt"
