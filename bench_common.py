"""Constants and synthetic-scene helpers shared by bench.py and its baseline / probe legs (bench_baselines.py, bench_probes.py):
the workload of BASELINE.json's metric -- sizes, peaks the roofline is priced against, seeded weights / poses / options."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "4d-facial-avatars_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

H = W = 512
N_COARSE, N_FINE = 64, 128
CHUNK = 65536
FLOP_PER_POINT = 1_100_032            # algorithmic forward FLOPs of the paper MLP per point
EXEC_FLOP_PER_POINT_F32 = 999_936     # FLOPs the exact-f32 kernel issues per point: the folded constant columns never enter the GEMMs
CHAIN_FLOP_PER_POINT = 918_784        # dX chain: 2 x (3*128 + 2*128*128 + 128*256 + 256 + 6*256*256)
DW_FLOP_PER_POINT = 1_100_032         # weight gradients: one outer product per weight = the forward's products
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X dense fp32 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0                 # MI355X HBM3E spec peak (MI355X_MICROARCH.md; about 6.3 TB/s is achievable)
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X dense bf16 MFMA (MI355X_MICROARCH.md; AMD's 5 PF figure is 2:1 sparse)
SPLIT_MFMAS_PER_TILE = {"x3": 2982, "x2": 1988}   # v_mfma_f32_32x32x16 per 32-point wave tile of the split inference kernels (round 5: 3012 / 2008)
BF16X3_EXEC_FLOP_PER_POINT = SPLIT_MFMAS_PER_TILE["x3"] * 32768 / 32   # executed MFMA FLOPs per point of the three-product kernels
INTRINSICS = np.array([-1481.96352, 1559.67488, 0.565694, 0.413902])
NEAR, FAR = 0.2, 0.8
CPU_CALIBRATION_RAYS = 4096           # slice of the CPU sample on which the thread count of the reference's CPU run is chosen


# algorithmic HBM bytes per MLP point of the three training kernels of the PAPER model (csrc/nf_mlp_layout.h): the forward writes the
# saved activations + ReLU bit masks (and reads z), the chain reads its ReLU masks + d_raw and writes dZ, the weight-gradient
# GEMMs read every saved activation, every dZ and d_raw once
TRAIN_KERNELS = {
    "f32": (("k_paper_mlp_fwd_save", "forward with saves"), ("k_paper_mlp_bwd_chain_masks", "dX chain"), ("k_dw_gemm_lds", "weight-gradient GEMMs")),
    "bf16x3": (("k_paper_mlp_fwd_bf16_train", "forward with saves"), ("k_paper_mlp_bwd_chain_bf16", "dX chain"), ("k_paper_dw_gemm_bf16", "weight-gradient GEMMs")),
    "f16x3": (("k_paper_mlp_fwd_f16_train", "forward with saves"), ("k_paper_mlp_bwd_chain_f16", "dX chain"), ("k_paper_dw_gemm_f16", "weight-gradient GEMMs")),
}
TRAIN_BYTES_PER_POINT = (4 * (2256 + 72) + 4 + 16, 4 * 72 + 16 + 4 * 2176, 4 * (2256 + 2176 + 4))
TRAIN_FLOP_PER_POINT = (FLOP_PER_POINT, CHAIN_FLOP_PER_POINT, DW_FLOP_PER_POINT)
# issued 16-bit MFMA FLOPs per point of the split training kernels: v_mfma_f32_32x32x16 = 32768 FLOPs per 32 points; the forward with
# saves issues 3254 per wave tile (2982 + 272 transposing ones), the dX chain 2760 (static counts of the ISA, tools/isa_summary.py),
# the weight-gradient GEMMs three products per algorithmic one
TRAIN_SPLIT_EXEC_FLOP_PER_POINT = (3254 * 1024, 2760 * 1024, 3 * DW_FLOP_PER_POINT)
# (lcode family, --mode train --family lcode: whole-iteration bytes only)
LCODE_BYTES_PER_POINT = {"f32": 4 * 1488 + (4 * (4 * 256 + 128) + 16) + 4 * 1408 + 4 * (1488 + 1408 + 4),
                         "bf16x3": 4 * 1528 + (40 + 16) + 4 * 1408 + 4 * (1488 + 1408 + 4),
                         "f16x3": 4 * 1528 + (40 + 16) + 4 * 1408 + 4 * (1488 + 1408 + 4)}


def synth_params(seed, device, family="paper"):
    """Random-init weights of the paper architecture (torch default nn.Linear init) with a density boost so
    that rays are neither all-empty nor all-opaque.  family="lcode": the second model family (--mode train only)."""
    import nerf
    torch.manual_seed(seed)
    cls = nerf.models.ConditionalBlendshapePaperNeRFModel if family == "paper" else nerf.models.ConditionalBlendshapeLearnableCodeNeRFModel
    m = cls(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=False, use_viewdirs=True,
            num_layers=4, hidden_size=256, include_expression=True)
    with torch.no_grad():
        m.fc_alpha.weight.mul_(1000.0)
        m.fc_alpha.bias.fill_(5.0)
        m.fc_rgb.weight.mul_(10.0)
    return m.to(device).eval()


def frame_pose(f):
    import math
    a = 0.3 * math.sin(2 * math.pi * f / 100.0)
    b = 0.15 * math.cos(2 * math.pi * f / 100.0)
    ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    m = np.eye(4)
    m[:3, :3] = ry @ rx
    m[:3, 3] = [0.02 * math.sin(2 * math.pi * f / 100.0), 0.02 * math.cos(2 * math.pi * f / 100.0), 0.5]
    return torch.tensor(m, dtype=torch.float32)


def options(nerf, chunk=CHUNK):
    mode = dict(num_coarse=N_COARSE, num_fine=N_FINE, chunksize=chunk, perturb=True, lindisp=False,
                radiance_field_noise_std=0.0, white_background=False)
    return nerf.CfgNode(dict(nerf=dict(use_viewdirs=True, train=dict(mode), validation=dict(mode)),
                             dataset=dict(no_ndc=True, near=NEAR, far=FAR)))
