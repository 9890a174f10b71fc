#!/usr/bin/env python3
"""Table of tools/r06_ring_bytes.sh's output: per block and message size, the graph's time per message under each byte budget.
   python tools/ring_bytes_table.py gpurun_out/r06_kpn_ring_bytes.txt"""
import json
import sys

budget = None
rows = {}
for line in open(sys.argv[1]):
    if line.startswith("budget"):
        budget = int(line.split()[1])
        continue
    if not line.startswith("{"):
        continue
    r = json.loads(line)
    if r["mode"] == "bench_block":
        key = (r["block"], r["msg_samples"])
    else:
        key = ("chain/" + r["source"] + ("/carried" if r.get("history") == "carried" else ""), 1 << r["log2_msg"])
    rows.setdefault(key, {}).setdefault(budget, []).append((r["bare_us_per_msg"], r["graph_us_per_msg"]))
budgets = sorted({b for v in rows.values() for b in v})
print("%-24s %10s %9s | graph us per message (bare / graph), one column per run, budget MiB: %s" % ("block", "samples", "bare us", " | ".join(map(str, budgets))))
for key in sorted(rows):
    bare = [b for v in rows[key].values() for b, _ in v]
    cols = [" ".join("%8.1f (%5.1f%%)" % (g, 100 * b / g) for b, g in rows[key].get(bd, [])) for bd in budgets]
    print("%-24s %10d %9.1f | %s" % (key[0], key[1], sum(bare) / len(bare), " | ".join(cols)))
