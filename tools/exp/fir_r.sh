python -m pytest tests/test_gpu_parity.py -x -q -k "fir" 2>&1 | tail -2
for s in "255 10" "101 3" "127 3" "200 4" "129 8" "77 5" "600 10" "31 3"; do python tools/shape_probe.py fir $s; python tools/shape_probe.py firx $s; done
python tools/shape_probe.py firr 101 3; python tools/shape_probe.py firr 255 10
