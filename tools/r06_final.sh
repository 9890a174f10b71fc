# Round-6 closing pass on the literal HEAD (one gpurun call): full GPU suite, smoke, the all-branch randomised run and the FAST / hard-resampler slices,
# the driver-flag bench line.   bash tools/r06_final.sh [TAG] [fuzz seconds]
TAG=${1:-r06h}; FS=${2:-420}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 2400 python3 -m pytest tests -m gpu -q > $O/${TAG}_pytest.txt 2>&1 ) 2> $O/${TAG}_pytest_time.txt; echo "pytest rc=$?" >> $O/${TAG}_pytest.txt
grep -E "passed|failed|rc=" $O/${TAG}_pytest.txt | tail -3; tail -3 $O/${TAG}_pytest_time.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1; tail -1 $O/${TAG}_smoke.txt
python3 tests/fuzz_parity.py $FS 60601 > $O/${TAG}_fuzz_all.txt 2>&1; grep -E "^runs|^FAIL" $O/${TAG}_fuzz_all.txt | tail -3
FUZZ_ONLY=13 python3 tests/fuzz_parity.py 120 60602 > $O/${TAG}_fuzz_fast.txt 2>&1; grep -E "^runs|^FAIL" $O/${TAG}_fuzz_fast.txt | tail -2
FUZZ_SRC_HARD=1 FUZZ_ONLY=5,9 python3 tests/fuzz_parity.py 120 60603 > $O/${TAG}_fuzz_srchard.txt 2>&1; grep -E "^runs|^FAIL" $O/${TAG}_fuzz_srchard.txt | tail -2
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_flags.json 2> $O/${TAG}_bench_driver_flags.err; cut -c1-300 $O/${TAG}_bench_driver_flags.json
