# round 5, call 6: full GPU suite (no -x: every flip-sensitive gate shows its value) on the branch-free sincos build; the split-products
# experiment (tools/split_products_probe.py: 9 variants of the split inference kernels)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c6; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v Warning > $O/pytest_gpu_full.txt; tail -8 $O/pytest_gpu_full.txt; grep -a "train step" $O/pytest_gpu_full.txt
timeout 1500 python tools/split_products_probe.py run > $O/split_products.txt 2>&1; grep "^|" $O/split_products.txt
