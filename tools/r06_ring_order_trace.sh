#!/bin/bash
# The order of the kernels in the GPU queue with and without the ring's byte budget (rocprofv3 --kernel-trace of `kpn_tests bench_block_list fft:24:400`):
# evidence for profiles/r06_kpn_ring_bytes.txt.   bash tools/r06_ring_order_trace.sh   (on the GPU box; writes gpurun_out/r06_ring_order_*.csv)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mib in 128 4194304; do
  export KPN_DEV_RING_MIB=$mib
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ringtrace_$mib -o t -- $R/tests/_build/kpn_tests bench_block_list fft:24:400 > $R/gpurun_out/r06_ring_order_$mib.log 2>&1
  f=$(find /tmp/ringtrace_$mib -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $mib <<'PY' > $R/gpurun_out/r06_ring_order_$mib.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = ["F" if "fft" in r["Kernel_Name"] else "c" if "checksum" in r["Kernel_Name"] else "." for r in rows]
dur = {"F": [], "c": []}
for r, n in zip(rows, names):
    if n in dur: dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
seq = "".join(names)
tail = seq[-240:]   # the end of the graph's timed region (the bare leg comes first in the process and alternates by construction)
print("budget MiB", sys.argv[2], "kernels", len(rows))
print("last 240 kernels of the graph leg (F = the 1024-point transform of one message, c = the sink's checksum of one message):")
print(tail)
# transforms that run between a message's transform and the checksum that reads it: the j-th last checksum belongs to the j-th last transform
fpos = [i for i, n in enumerate(names) if n == "F"]
cpos = [i for i, n in enumerate(names) if n == "c"]
gaps = []
for j in range(1, 301):
    f, c = fpos[-j], cpos[-j]
    gaps.append(sum(1 for q in fpos[-j:] if f < q < c))
import collections
print("transforms between a message's transform and its checksum, last 300 messages:", dict(sorted(collections.Counter(gaps).items())))
half = len(dur["c"]) // 2
med = lambda v: sorted(v)[len(v) // 2]
print("median kernel us, graph leg (second half of the trace): transform %.1f, checksum %.1f" % (med(dur["F"][len(dur["F"]) // 2:]), med(dur["c"][half:])))
PY
done
