# round 6: the whole GPU suite with timing of the slowest tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
( time timeout 2400 python3 -m pytest tests -m gpu -q --durations=25 > $O/r06_pytest.txt 2>&1 ) 2> $O/r06_pytest_time.txt; echo "pytest rc=$?" >> $O/r06_pytest.txt
grep -E "passed|failed|rc=|^FAILED|^ERROR" $O/r06_pytest.txt | tail -15; cat $O/r06_pytest_time.txt
grep -A30 "slowest" $O/r06_pytest.txt | head -32
