# round 6: the driver-flag bench line with the new legs, then the moved-bytes PMC passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_kpn_cpp.py tests/test_gpu_runs.py -m gpu -q -x > $O/r06_kpn_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r06_kpn_pytest.txt
tail -3 $O/r06_kpn_pytest.txt
( time python3 bench.py --steps 20 --warmup 5 > $O/r06a_bench_driver_flags.json 2> $O/r06a_bench_driver_flags.err ) 2> $O/r06a_bench_time.txt
tail -3 $O/r06a_bench_time.txt; tail -3 $O/r06a_bench_driver_flags.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r06a_bench_driver_flags.json'))
print(d['value'], d['roofline']['frac'])
o=d['other_configs']
for k,v in o.items():
    if isinstance(v,dict) and 'kernel_ms' in v: print(k, v['kernel_ms'], v['frac'], v.get('moved_bytes_over_alg'), v.get('cpu_baseline',{}).get('value'), v.get('cpu_baseline',{}).get('single_core'))
print(json.dumps(o['kpn_graph_c2'], indent=0)[:3000])
print(json.dumps(o['dropin_calls'], indent=0))
print(d['from_u8_bytes'])
PY
bash tools/pmc_moved.sh 06 > $O/r06_pmc_moved.txt 2>&1; tail -40 $O/r06_pmc_moved.txt
