# Round-5 closing pass on the literal HEAD (one gpurun call): full GPU suite, smoke, the all-branch randomised run (>= 300 s) and the FAST branch,
# the driver-flag bench line, the C3 / C4 / FIR lines.   bash tools/r05_final.sh [TAG]
TAG=${1:-r05h}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -m gpu -q > $O/${TAG}_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/${TAG}_pytest.txt
grep -E "passed|failed|rc=" $O/${TAG}_pytest.txt | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.txt 2>&1; tail -1 $O/${TAG}_smoke.txt
python3 tests/fuzz_parity.py 480 50501 > $O/${TAG}_fuzz_all.txt 2>&1; grep -E "^runs|^FAIL" $O/${TAG}_fuzz_all.txt | tail -3
FUZZ_ONLY=13 python3 tests/fuzz_parity.py 150 50502 > $O/${TAG}_fuzz_fast.txt 2>&1; grep -E "^runs|^FAIL" $O/${TAG}_fuzz_fast.txt | tail -2
FUZZ_SRC_HARD=1 FUZZ_ONLY=5,9 python3 tests/fuzz_parity.py 150 50503 > $O/${TAG}_fuzz_srchard.txt 2>&1; grep -E "^runs|^FAIL" $O/${TAG}_fuzz_srchard.txt | tail -2
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_flags.json 2> $O/${TAG}_bench_driver_flags.err; cut -c1-300 $O/${TAG}_bench_driver_flags.json
python3 tools/bench_configs.py c3 c3big c4 firshapes srcsmall > $O/${TAG}_lines.txt 2>&1; grep -v amdgpu $O/${TAG}_lines.txt | cut -c1-170
