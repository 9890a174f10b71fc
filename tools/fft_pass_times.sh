# per-kernel times of the multi-pass FFT sizes (gpurun, from the repository root):  bash tools/fft_pass_times.sh TAG
TAG=${1:-r02_fftpass}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lg in ${LGS:-15 16 17 18 19 20 21 22 23 24}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_$lg -- python3 $R/tools/shape_probe.py fft $((1<<lg)) 0 26 10 > $O/${TAG}_$lg.txt 2>&1
  f=$(find $O/${TAG}_$lg -name "*kernel_stats.csv" | head -1)
  echo "== 2^$lg: $(tail -1 $O/${TAG}_$lg.txt)"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "fft" in n or "ovsave" in n:
        print(f'   {n[:70]:70s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
done > $O/${TAG}_summary.txt 2>&1
cat $O/${TAG}_summary.txt
find $O -name "*kernel_trace.csv" -size +1M -delete
