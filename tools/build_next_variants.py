"""Experiment builds waiting for their first GPU run (written at the end of round 3 with the GPU budget spent):
    prefetch      -DNF_FWD_PREFETCH_IN=1   f32 inference kernel: the next block's z / ray loads issued under the dir layers (persistent loop)
    lcode_stream  -DNF_LCODE_STREAM=1      second model family, f32 inference: layer-streamed body
Build here (the .so files travel with the gpurun snapshot), then:  gpurun -- 'bash tools/gpu_calls/next_round_ab.sh'"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "4d-facial-avatars_amd"))
import build  # noqa: E402

build.build(verbose=False)
for suffix, flags in (("prefetch", ["-DNF_FWD_PREFETCH_IN=1"]), ("lcode_stream", ["-DNF_LCODE_STREAM=1"])):
    print(build.build_variant(suffix, flags, verbose=False))
