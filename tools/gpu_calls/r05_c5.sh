# round 5, call 5: branch-free in-kernel sincos, rotated coordinate selection (headline kernel: 38 -> 0 spilled SGPRs), k_grad_reduce tail
# without dynamic accumulator index (29 -> 0): full GPU suite + kernel times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | tail -15 > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
timeout 300 python tools/time_mlp.py > $O/time_mlp.txt 2>&1; tail -12 $O/time_mlp.txt
timeout 300 python tools/time_train_split.py bf16x3 f16x3 f32 > $O/train.txt 2>&1; tail -4 $O/train.txt
