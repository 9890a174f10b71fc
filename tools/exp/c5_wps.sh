for i in 1 2; do
python tools/bench_configs.py c5only 2>&1 | grep "C5 "
REDIO_BUILD_DIR=$PWD/tools/exp/_build_wps3 python tools/bench_configs.py c5only 2>&1 | grep "C5 "
done
REDIO_BUILD_DIR=$PWD/tools/exp/_build_wps3 python -m pytest tests/test_gpu_overlap_save.py -x -q 2>&1 | tail -2
