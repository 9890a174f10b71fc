# A/B of one measurement script between the product build and an older build of the library, alternating PROCESSES on one box (gpurun):
#   here:   git stash (or checkout the old sources); make -C libredio_amd/csrc OUT=../../tools/exp/_build_old; restore the sources; make -C libredio_amd/csrc
#   box:    bash tools/ab_old_build.sh tools/c4_time.py [rounds]        (round 5: the channelizer's waits, the chain's late stores)
S=${1:-tools/c4_time.py}; N=${2:-2}
cd $GRAFT_REPO_ROOT
for i in $(seq $N); do
  echo "== product build"; python3 $S 2>&1 | grep -v amdgpu.ids
  echo "== old build (tools/exp/_build_old)"; REDIO_BUILD_DIR=$GRAFT_REPO_ROOT/tools/exp/_build_old python3 $S 2>&1 | grep -v amdgpu.ids
done
