# round 4, call 9: which clock do the training kernels hold?  (busy cycles / duration per dispatch)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c9; mkdir -p $O
timeout 300 python tools/kernel_clocks.py pmc_train_launch.py bf16x3 f16x3 f32 2>&1 | tail -20 | tee $O/train_clocks.md
timeout 300 python tools/kernel_clocks.py pmc_one_launch.py bf16x3 2>&1 | tail -5 | tee $O/infer_bf16_clocks.md
timeout 300 python tools/kernel_clocks.py pmc_one_launch.py f32 2>&1 | tail -5 | tee $O/infer_f32_clocks.md
