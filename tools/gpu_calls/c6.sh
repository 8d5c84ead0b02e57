# GPU call 6 (round 3): store cache-policy variants of the slab copy; PMC of the three f32 training kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" _nt _sc1 _ntsc ""; do
  echo "== variant '$v'"
  NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 120 python tools/time_train_f32.py 10 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c6/variants.log 2>&1
cat gpurun_out/c6/variants.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA -d /tmp/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_train_launch.py f32 > $GRAFT_REPO_ROOT/gpurun_out/c6/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM -d /tmp/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_train_launch.py f32 > $GRAFT_REPO_ROOT/gpurun_out/c6/pmc2.log 2>&1; echo "pmc2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/pmc3 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_train_launch.py f32 > $GRAFT_REPO_ROOT/gpurun_out/c6/pmc3.log 2>&1; echo "pmc3 rc=$?"
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py pmc $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 -name "*.db") > $GRAFT_REPO_ROOT/gpurun_out/c6/pmc.md 2>&1
grep "k_paper_mlp_fwd_save\|chain_masks\|dw_gemm_lds" $GRAFT_REPO_ROOT/gpurun_out/c6/pmc.md | cut -c20-150
tail -3 $GRAFT_REPO_ROOT/gpurun_out/c6/pmc2.log
