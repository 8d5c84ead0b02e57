# GPU call 3 (round 3): same-box A/B of the K-loop variants + ablations of the exact-f32 eval kernel, dW with a 1-D grid, PMC of the eval kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" _r02loop _nofence _nowload _noepi _nope ""; do
  echo "== variant '$v'"
  NERFACE_HIP_LIB=$L/libnerface_hip$v.so TIME_MLP_ONLY_F32=1 timeout 120 python tools/time_mlp.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c3/variants.log 2>&1
cat gpurun_out/c3/variants.log
for v in "" _r02loop; do
  echo "== train A/B with variant '$v'"
  NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/ab_train_f32.py --json gpurun_out/c3/ab$v.json 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c3/ab.log 2>&1
cat gpurun_out/c3/ab.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o ab -- python $GRAFT_REPO_ROOT/tools/ab_train_f32.py --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/c3/prof.log 2>&1; echo "prof rc=$?"
DB=$(find /tmp/prof_ab -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $DB > $GRAFT_REPO_ROOT/gpurun_out/c3/ab_kernel_stats.md 2>&1
head -12 $GRAFT_REPO_ROOT/gpurun_out/c3/ab_kernel_stats.md | cut -c1-150
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d /tmp/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one_launch.py f32 > $GRAFT_REPO_ROOT/gpurun_out/c3/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES -d /tmp/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one_launch.py f32 > $GRAFT_REPO_ROOT/gpurun_out/c3/pmc2.log 2>&1; echo "pmc2 rc=$?"
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py pmc $(find /tmp/pmc1 /tmp/pmc2 -name "*.db") > $GRAFT_REPO_ROOT/gpurun_out/c3/pmc.md 2>&1
grep "k_paper_mlp_fwd" $GRAFT_REPO_ROOT/gpurun_out/c3/pmc.md | cut -c1-160
