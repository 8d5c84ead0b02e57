import sys, time, torch
sys.path.insert(0, "4d-facial-avatars_amd"); sys.path.insert(0, ".")
import nerf
import math
import numpy as np
dev = torch.device("cuda:0")
m = nerf.models.ConditionalBlendshapeLearnableCodeNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=False, use_viewdirs=True, num_layers=4, hidden_size=256, include_expression=True)
def boost(mm):      # density / colour heads scaled so that rays are neither all-empty nor all-opaque (as bench.py does)
    with torch.no_grad():
        mm.fc_alpha.weight.mul_(300.0); mm.fc_alpha.bias.fill_(5.0); mm.fc_rgb.weight.mul_(10.0)
    return mm
torch.manual_seed(6); m = boost(m).to(dev)
R, S = 65536, 192
ro = torch.zeros(R, 3, device=dev); rd = torch.randn(R, 3, device=dev) * 0.3; z = torch.sort(torch.rand(R, S, device=dev) * 0.6 + 0.2, dim=-1)[0]
expr = torch.randn(76, device=dev); lat = torch.randn(32, device=dev) * 0.1
for prec in ("f32", "bf16x3", "f16x3", "f16x2"):
    nerf.set_mlp_precision(prec)
    for _ in range(2): m.hip_forward(ro, rd, z, rd, expr, lat, 0.2, 0.8, False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): m.hip_forward(ro, rd, z, rd, expr, lat, 0.2, 0.8, False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(f"lcode {prec} {R}x{S}: {dt*1e3:.2f} ms  {R*S*684800/dt/1e12:.1f} TFLOP/s algorithmic")

# whole 512x512 frame (64 + 128 samples, chunksize 65536) through run_one_iter_of_nerf, second family
import numpy as np
mc = nerf.models.ConditionalBlendshapeLearnableCodeNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=False, use_viewdirs=True, num_layers=4, hidden_size=256, include_expression=True)
torch.manual_seed(5); mc = boost(mc).to(dev)
mode = dict(num_coarse=64, num_fine=128, chunksize=65536, perturb=True, lindisp=False, radiance_field_noise_std=0.0, white_background=False)
opt = nerf.CfgNode(dict(nerf=dict(use_viewdirs=True, train=dict(mode), validation=dict(mode)), dataset=dict(no_ndc=True, near=0.2, far=0.8)))
ex = nerf.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True)
ed = nerf.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True)
bg = torch.rand(512 * 512, 3, device=dev)
INTRINSICS = np.array([-1481.96352, 1559.67488, 0.565694, 0.413902])
_a = 0.3 * math.sin(2 * math.pi * 3 / 100.0)
POSE = torch.tensor([[math.cos(_a), 0, math.sin(_a), 0.0], [0, 1, 0, 0.02], [-math.sin(_a), 0, math.cos(_a), 0.5], [0, 0, 0, 1]], dtype=torch.float32, device=dev)
def frame():
    ro, rd = nerf.get_ray_bundle(512, 512, INTRINSICS, POSE)
    with torch.no_grad():
        return nerf.run_one_iter_of_nerf(512, 512, INTRINSICS, mc, m, ro, rd, opt, mode="validation", encode_position_fn=ex,
                                         encode_direction_fn=ed, expressions=expr, background_prior=bg, latent_code=lat)
for prec in ("f32", "bf16x3", "f16x3", "f16x2"):
    nerf.set_mlp_precision(prec)
    for _ in range(2): frame()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3): frame()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    print(f"lcode {prec} full frame 512x512, 64+128: {dt*1e3:.1f} ms = {512*512/dt/1e6:.2f} M rays/s")
