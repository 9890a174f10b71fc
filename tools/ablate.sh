# Timing-only ablation builds of the chain kernel (results are WRONG by construction; never shipped, never tested
# for parity): what share of the kernel's time and power goes to the LDS window reads and to the multiply-adds?
#   build here:   bash tools/ablate.sh build         (cross-compiles tools/exp/_build_abl{1,2}/libredio.so)
#   on the box:   bash tools/ablate.sh run            (power_probe on the product build and on both ablations)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = "build" ]; then
  for a in 1 2; do
    mkdir -p $R/tools/exp/_build_abl$a
    make -C $R/libredio_amd/csrc -s -j8 OUT=$R/tools/exp/_build_abl$a EXTRA=-DREDIO_EXP_ABLATE=$a $R/tools/exp/_build_abl$a/libredio.so
  done
else
  cd $R
  echo "== product build"; python3 tools/power_probe.py ${2:-1500} fused
  echo "== ablation 1: half the LDS window reads (same multiply-adds, same HBM bytes)"; REDIO_BUILD_DIR=$R/tools/exp/_build_abl1 python3 tools/power_probe.py ${2:-1500} fused
  echo "== ablation 2: half the multiply-adds (same LDS reads, same HBM bytes)"; REDIO_BUILD_DIR=$R/tools/exp/_build_abl2 python3 tools/power_probe.py ${2:-1500} fused
fi
