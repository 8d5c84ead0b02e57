# round 4, call 19: k_resample_merge_small: branch-free descent; stage costs by early-exit variants (timing only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c19; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -k "resample or sample_pdf or sort" 2>&1 | grep -v Warning | tail -25 > $O/pytest.txt; cat $O/pytest.txt | tail -25
for v in _prev "" _rs1 _rs2 _rs3 _rs4 ""; do echo "== lib${v:-_new}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/resample_check.py 2>&1 | grep "resample\|Error" | head -4; done > $O/resample_ab.txt; cat $O/resample_ab.txt
