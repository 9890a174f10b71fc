# SQ issue/stall counters for one plan shape (gpurun, from the repository root):
#   bash tools/shape_pmc.sh TAG KERNEL_SUBSTRING fir 64 1
TAG=$1; KSUB=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/shape_probe.py "$@" > $O/${TAG}_time.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/${TAG}_sq1 -- python3 $R/tools/shape_probe.py "$@" > /dev/null 2> $O/${TAG}_sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${TAG}_sq2 -- python3 $R/tools/shape_probe.py "$@" > /dev/null 2> $O/${TAG}_sq2.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_sq3 -- python3 $R/tools/shape_probe.py "$@" > /dev/null 2> $O/${TAG}_sq3.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_fetch -- python3 $R/tools/shape_probe.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_write -- python3 $R/tools/shape_probe.py "$@" > /dev/null 2>&1
cd $R
{ echo "# $KSUB ($*), mean per launch"; cat $O/${TAG}_time.txt; python3 profiles/pmc_summary.py "$KSUB" $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_sq1 $O/${TAG}_sq2 $O/${TAG}_sq3; } > $O/${TAG}_pmc_summary.txt 2>&1
cat $O/${TAG}_pmc_summary.txt
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
