"""Fingerprint of the second family's exact-f32 training step (raw, saved activations, gradients) and its iteration-kernel time for the
library NERFACE_HIP_LIB selects: a restructured kernel must leave every hash unchanged."""
import argparse, hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
dev = torch.device("cuda:0")
nerf.set_mlp_precision("f32")
m = bench.synth_params(1, dev, "lcode").train()
g = torch.Generator().manual_seed(11)
expr, lat = (torch.randn(76, generator=g) * 0.5).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev)
sha = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:12]
for n_rays, s in ((3, 7), (37, 128), (2047, 127), (2048, 128)):
    ro = torch.zeros(n_rays, 3, device=dev); rd = (torch.randn(n_rays, 3, generator=g) * 0.3).to(dev)
    z = torch.sort(torch.rand(n_rays, s, generator=g) * 0.6 + 0.2, dim=-1)[0].to(dev)
    d_raw = (torch.randn(n_rays, s, 4, generator=g) / (3 * n_rays)).to(dev)
    raw, state = m.hip_forward(ro, rd, z, rd, expr, lat, 0.2, 0.8, True)
    n = n_rays * s
    saved = state[2][:1528 * n].clone()
    grads, g_lat = m.hip_backward(state, z, d_raw)
    torch.cuda.synchronize()
    hg = hashlib.sha1(b"".join(x.cpu().numpy().tobytes() for x in list(grads) + [g_lat] if x is not None)).hexdigest()[:12]
    print(f"lcode {n_rays}x{s}: raw {sha(raw)} saved {sha(saved)} grads {hg}", flush=True)
for _ in range(2):
    r = bench.train_roofline(argparse.Namespace(precision="f32", family="lcode"), m, dev, 2048)
print(f"lcode f32: MLP kernels of an iteration {r['ms_both_launches']:.3f} ms", flush=True)
