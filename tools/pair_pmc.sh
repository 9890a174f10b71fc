# Kernel times and SQ / LDS counters of the four-stage tile passes in the two builds -- the two-column "pair" program (product build)
# and the one-column program (make OUT=../_build_onecol EXTRA=-DREDIO_TILE_PAIR=0) -- side by side.  On the GPU box:
#   bash tools/pair_pmc.sh [TAG]      -> gpurun_out/TAG_summary.txt
TAG=${1:-pairpmc}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/${TAG}_counters_avail.txt 2>&1
for B in pair onecol; do
  if [ $B = onecol ]; then export REDIO_BUILD_DIR=$R/libredio_amd/_build_onecol; else unset REDIO_BUILD_DIR; fi
  rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_${B}_kt -- python3 $R/tools/pair_probe.py 30 > $O/${TAG}_${B}_kt.out 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/${TAG}_${B}_sq1 -- python3 $R/tools/pair_probe.py 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM --output-format csv -d $O/${TAG}_${B}_sq2 -- python3 $R/tools/pair_probe.py 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_IFETCH GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_${B}_sq3 -- python3 $R/tools/pair_probe.py 6 > /dev/null 2> $O/${TAG}_${B}_sq3.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_${B}_fetch -- python3 $R/tools/pair_probe.py 6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_${B}_write -- python3 $R/tools/pair_probe.py 6 > /dev/null 2>&1
done
unset REDIO_BUILD_DIR
cd $R
{
for B in pair onecol; do
  echo "######## build: $B"
  python3 - $O/${TAG}_${B}_kt <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    if "synth" in k: continue
    print(f"{k:72s} n={len(v):4d} mean={sum(v)/len(v):9.1f} us  median={v2[len(v2)//2]:9.1f}  min={v2[0]:9.1f}")
PY
  for k in "fftbig_first_kernel<false>" "fftbig_mid_kernel<false>" "ovsave64k_step_kernel"; do echo "== $k"; python3 profiles/pmc_summary.py "$k" $O/${TAG}_${B}_sq1 $O/${TAG}_${B}_sq2 $O/${TAG}_${B}_sq3 $O/${TAG}_${B}_fetch $O/${TAG}_${B}_write; done
done
} > $O/${TAG}_summary.txt 2>&1
find $O -name "*kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
cat $O/${TAG}_summary.txt
