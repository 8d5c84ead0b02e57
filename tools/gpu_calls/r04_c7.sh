# round 4, call 7: where do the split training kernels wait?  SQ counters of the bf16x3 training kernels and of the bf16x3 inference kernel
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_c7; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA -d /tmp/pmc1 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_train_launch.py bf16x3 > $O/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM -d /tmp/pmc2 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_train_launch.py bf16x3 > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC -d /tmp/pmc3 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_train_launch.py bf16x3 > $O/pmc3.log 2>&1; echo "pmc3 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA -d /tmp/pmc4 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one_launch.py bf16x3 > $O/pmc4.log 2>&1; echo "pmc4 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/pmc5 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_one_launch.py bf16x3 > $O/pmc5.log 2>&1; echo "pmc5 rc=$?"
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py pmc $(find /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 /tmp/pmc4 /tmp/pmc5 -name "*.db") > $O/pmc.md 2>&1
grep "k_paper_mlp_fwd_bf16\|chain_bf16\|dw_gemm_bf16" $O/pmc.md | cut -c18-150
tail -2 $O/pmc3.log
