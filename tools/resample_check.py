"""k_resample_merge_small: fingerprints and HIP-event time for the library NERFACE_HIP_LIB selects (65536 x (64 + 128) as in an eval chunk,
2048 x (64 + 64) as in a training iteration; random and deterministic abscissae)."""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch
from nerf import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
sha = lambda *ts: hashlib.sha1(b"".join(t.detach().cpu().numpy().tobytes() for t in ts)).hexdigest()[:12]
for n_rays, nc, nf in ((65536, 64, 128), (2048, 64, 64), (4099, 128, 128), (777, 33, 100)):
    z = torch.sort(torch.rand((n_rays, nc), generator=g) * 0.6 + 0.2, dim=-1)[0].to(dev)
    w = (torch.rand((n_rays, nc), generator=g) ** 6).to(dev)
    u = torch.rand((n_rays, nf), generator=g).to(dev)
    for name, uu in (("random u", u), ("linspace", None)):
        zf, zs = ops.resample_merge(z, w, nf, uu, want_samples=True)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ops.resample_merge(z, w, nf, uu)
        a.record()
        for _ in range(50):
            ops.resample_merge(z, w, nf, uu)
        b.record(); torch.cuda.synchronize()
        print(f"resample_merge {n_rays} x ({nc} + {nf}) {name}: z_fine {sha(zf)} z_samples {sha(zs)}  {a.elapsed_time(b) / 50 * 1e3:.1f} us per call (incl. launch + allocation)", flush=True)
