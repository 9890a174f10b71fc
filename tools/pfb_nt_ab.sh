# A/B of the channelizer kernels under each cache policy: builds of the whole library per policy (make -C libredio_amd/csrc OUT=../../tools/exp/_build_pfbntN EXTRA=-DREDIO_EXP_PFB_NT=N
# with the pfb64 policy still a macro at the time; today pfb64's policy is a template parameter: tools/c4_nt_sweep.py), alternating processes on one box
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo "== product build (non-temporal row loads and stores)"; python3 tools/c4gen_time.py 2>&1 | grep -v amdgpu.ids
  for v in 0 1 2; do echo "== REDIO_EXP_PFB_NT=$v (0: default policy, 1: nt loads only, 2: nt stores only)"; REDIO_BUILD_DIR=$GRAFT_REPO_ROOT/tools/exp/_build_pfbnt$v python3 tools/c4gen_time.py 2>&1 | grep -v amdgpu.ids; done
done
