# per-pass durations of the large power-of-two transforms (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_big -- python3 $R/tools/bench_bigfft.py > $R/gpurun_out/p_big.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/p_big/*/*kernel_trace.csv'))[-1]
import collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'fftbig' in n or 'fft64k' in n:
        acc[(n.split('(')[0][-44:], r['Grid_Size'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: (int(kv[0][1]), kv[0][0])):
    print(f"{k[0]:46s} grid {k[1]:>10s} n={len(v):3d} mean {sum(v)/len(v):8.1f} us")
PY
