python -m pytest tests/test_gpu_parity.py -x -q -k "fir" 2>&1 | tail -2
for s in "255 10" "101 3" "127 3" "200 4" "129 8" "77 5" "600 10" "31 3" "64 1" "31 2" "500 2"; do python tools/shape_probe.py fir $s; python tools/shape_probe.py firx $s; done
