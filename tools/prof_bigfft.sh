# per-pass durations of the large power-of-two transforms (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_big -- python3 $R/tools/bench_bigfft.py > $R/gpurun_out/p_big.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/p_big/*/*kernel_trace.csv'))[-1]
rows = [r for r in csv.DictReader(open(f)) if 'fftbig' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seqs, cur = [], []
for r in rows:  # a gather pass starts a new transform
    n = r['Kernel_Name'].split('(')[0].replace('void redio::', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'first' in n and cur: seqs.append(cur); cur = []
    cur.append((n, d))
seqs.append(cur)
groups = []
for sq in seqs:
    key = tuple(n for n, _ in sq)
    if groups and groups[-1][0] == key: groups[-1][1].append(sq)
    else: groups.append((key, [sq]))
for key, lst in groups:
    use = lst[2:] or lst
    means = [sum(sq[i][1] for sq in use) / len(use) for i in range(len(key))]
    print(' + '.join(f"{k.replace('fftbig_', '').replace('_kernel', '')}:{m:.0f}" for k, m in zip(key, means)), f"= {sum(means):.0f} us")
PY
