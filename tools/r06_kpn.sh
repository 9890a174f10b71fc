# round 6: the device-resident kpn graph (include/kpn_dev.hpp rings + queue order): tests, then the bench_c2 sweep (shipped / per-block streams / round-5 lines)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -f $O/r06_kpn_*.txt
timeout 1200 python3 -m pytest tests/test_kpn_cpp.py tests/test_gpu_runs.py -m gpu -q -x > $O/r06_kpn_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r06_kpn_pytest.txt
tail -5 $O/r06_kpn_pytest.txt
timeout 600 tests/_build/kpn_tests bench_c2_sweep 0.3 3 16 28 > $O/r06_kpn_sweep.txt 2>&1; echo "sweep rc=$?" >> $O/r06_kpn_sweep.txt
cat $O/r06_kpn_sweep.txt
for src in resident synth; do for snk in checksum drop; do
timeout 120 tests/_build/kpn_tests bench_c2 24 2000 4 $src $snk 0 0 >> $O/r06_kpn_variants.txt 2>&1
timeout 120 tests/_build/kpn_tests bench_c2 28 100 4 $src $snk 0 0 >> $O/r06_kpn_variants.txt 2>&1
done; done
cat $O/r06_kpn_variants.txt
