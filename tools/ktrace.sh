# kernel durations of one command under rocprofv3 (gpurun, from the repository root):  bash tools/ktrace.sh TAG python3 tools/shape_probe.py ...
# (the command runs from /tmp: script arguments that are repository-relative paths are rewritten to $GRAFT_REPO_ROOT/...)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
ARGS=()
for a in "$@"; do
  if [ "${a#/}" = "$a" ] && [ -e "$R/$a" ]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_kt -- "${ARGS[@]}" > $O/${TAG}_kt.out 2>&1
cd $R
if [ -z "$(find $O/${TAG}_kt -name '*kernel_trace.csv' 2>/dev/null | head -1)" ]; then
  echo "ktrace.sh: no kernel trace was produced; the command's output:" >&2; tail -20 $O/${TAG}_kt.out >&2; exit 1
fi
python3 - $O/${TAG}_kt <<'PY' | tee $O/${TAG}_kt_summary.txt
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k:92s} n={len(v):4d} mean={sum(v)/len(v):10.1f} us  median={v2[len(v2)//2]:10.1f}  min={v2[0]:10.1f}")
PY
find $O/${TAG}_kt -name "*.csv" -size +2M -delete
