# round 4, call 1: full GPU suite on the default library (new: unmodified TR/EV scripts, --as-shipped, Adam resume), the second
# family's streamed inference variant (full lcode suite + eval timing A/B), then the whole bench line (summary tail, reference CPU leg)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c1; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
NERFACE_HIP_LIB=$L/libnerface_hip_lcode_stream.so timeout 600 python -m pytest tests/test_gpu_lcode.py -q -m gpu 2>&1 | tail -4 > $O/lcode_stream_pytest.txt; tail -2 $O/lcode_stream_pytest.txt
for v in "" lcode_stream "" lcode_stream; do echo "== ${v:-default}"; NERFACE_HIP_LIB=$L/libnerface_hip${v:+_$v}.so timeout 300 python tools/time_lcode.py 2>&1 | grep "lcode f32"; done > $O/lcode_ab.txt; cat $O/lcode_ab.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
