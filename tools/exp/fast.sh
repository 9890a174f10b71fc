python -m pytest tests/test_gpu_resample.py -x -q 2>&1 | tail -2
python tools/bench_configs.py c3 2>&1 | grep "fast f32"
python tools/bench_configs.py c3 2>&1 | grep "fast f32"
cd /tmp; bash $GRAFT_REPO_ROOT/tools/ktrace.sh fastp python3 $GRAFT_REPO_ROOT/tools/shape_probe.py srcfast 256 20 2>&1 | grep -v amdgpu.ids | head -3
