"""rocprofv3 counter_collection.csv of tools/pmc_moved.sh -> profiles/rNN_moved_bytes.json: HBM bytes per CALL of the multi-pass configs,
summed over every kernel the calls dispatch (the probe's input generator excluded); `prefix` names the dominant kernel family, which
bench.py checks against the kernels it launches.  FETCH_SIZE is doubled (the gfx950 correction for 16 B/lane
streaming reads, MI355X_MICROARCH.md HBM/rocprofv3 section), both counters are in KiB.
usage: pmc_moved.py OUT.json ROUND key:prefix:ncalls:fetch_dir:write_dir ..."""
import csv, glob, json, sys, collections

out, rnd, specs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
res = {"round": rnd, "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of tools/moved_probe.py (exactly `calls` calls of the "
                             "config at bench.py's other_configs size, nothing else on the device); bytes_per_call = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over "
                             "the call's kernels / calls"}


def total(d, counter, prefix):
    tot, names = 0.0, collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("redio::", "")
            if r["Counter_Name"] == counter and not name.startswith("synth_"):  # every kernel of the calls; the probe's input generator is not one
                tot += float(r["Counter_Value"]); names[name] += 1
    return tot, names


for sp in specs:
    key, prefix, ncalls, fd, wd = sp.split(":")
    ncalls = int(ncalls)
    f, fn = total(fd, "FETCH_SIZE", prefix)
    w, wn = total(wd, "WRITE_SIZE", prefix)
    res[key] = {"bytes_per_call": (2.0 * f + w) * 1024.0 / ncalls, "fetch_size_kb_per_call": f / ncalls, "write_size_kb_per_call": w / ncalls,
                "kernels": sorted(fn), "dominant_prefix": prefix, "dispatches_per_call": {k: v / ncalls for k, v in sorted(fn.items())}, "calls": ncalls}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
