# HBM bytes one call of the multi-pass configs moves (bench.py other_configs' moved_bytes_over_alg): rocprofv3 PMC passes, FETCH_SIZE and
# WRITE_SIZE apart (run on the GPU box: bash tools/pmc_moved.sh [ROUND]) -> gpurun_out/rNN_moved_bytes.json (copy to profiles/)
RND=${1:-06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in c5 fft65536; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/mv_${w}_fetch -- python3 $R/tools/moved_probe.py $w 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/mv_${w}_write -- python3 $R/tools/moved_probe.py $w 6 > /dev/null 2>&1
done
cd $R
python3 tools/pmc_moved.py $O/r${RND}_moved_bytes.json $RND c5_overlap_save_65536:ovsave64k:6:$O/mv_c5_fetch:$O/mv_c5_write fft_65536:fftbig:6:$O/mv_fft65536_fetch:$O/mv_fft65536_write
find $O -name "*counter_collection.csv" -size +8M -delete
find $O -name "*kernel_trace.csv" -size +2M -delete
