"""Fingerprints of the two exact-f32 kernels ported to the streamed K loops last (k_tiny_mlp_fwd, k_paper_mlp_fwd_encoded) for the
library NERFACE_HIP_LIB selects: a restructured kernel must leave every hash unchanged.  Also their kernel times (HIP events)."""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
import tiny_nerf as TN
dev = torch.device("cuda:0")
nerf.set_mlp_precision("f32")
sha = lambda *ts: hashlib.sha1(b"".join(t.detach().cpu().numpy().tobytes() for t in ts)).hexdigest()[:12]


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


# ---- tiny: inference, differentiable forward, gradients -------------------------------------------------------------------------
g = torch.Generator().manual_seed(5)
model = TN.VeryTinyNerfModel(num_encoding_functions=10).to(dev)
with torch.no_grad():
    for p in model.parameters():
        p.copy_((torch.randn(p.shape, generator=g) * (0.3 if p.dim() == 1 else (2.0 / p.shape[-1]) ** 0.5)).to(dev))
for hw in (5, 64, 100):
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.3, -0.2, 4.0])
    focal = torch.tensor(138.88 * hw / 100.0)
    call = lambda: TN.run_one_iter_of_tinynerf(hw, hw, focal, pose.to(dev), 2.0, 6.0, 32, None, TN.get_minibatches, 16384, model, 10)
    torch.manual_seed(3)
    with torch.no_grad():
        rgb_inf = call()
    torch.manual_seed(3)
    rgb = call()
    for p in model.parameters():
        p.grad = None
    (rgb * torch.linspace(0.5, 1.5, rgb.numel(), device=dev).reshape(rgb.shape)).sum().backward()
    print(f"tiny {hw}x{hw}x32: inference {sha(rgb_inf)} training forward {sha(rgb)} grads {sha(*[p.grad for p in model.parameters()])}", flush=True)
with torch.no_grad():
    ms_inf = timed(call)
print(f"tiny 100x100x32: inference iteration {ms_inf * 1e3:.1f} us (whole call, launch-bound)", flush=True)

# ---- paper model on pre-encoded inputs ---------------------------------------------------------------------------------------------
m = bench.synth_params(1, dev, "paper").eval()
expr, lat = (torch.randn(76, generator=g) * 0.5).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev)
for n in (1, 33, 4097, 262144):
    x87 = (torch.randn(n, 87, generator=g)).clamp(-1, 1).to(dev)
    with torch.no_grad():
        out = m(x87, expr, lat)
    print(f"encoded forward n={n}: {sha(out)}", flush=True)
with torch.no_grad():
    ms = timed(lambda: m(x87, expr, lat), 10)
print(f"encoded forward n=262144: {ms:.3f} ms per call", flush=True)
