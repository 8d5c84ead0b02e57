"""Informational: the reference ALGORITHM as stock PyTorch-ROCm eager ops on the same MI355X (the oracle's torch
restatement moved to cuda, chunked like the reference: 65536-ray chunks, 65536-point MLP chunks would need >4 GB per
concat, so rays are fed 8192 at a time).  Not part of the product or of bench.py; prints rays/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cases as C, nerface_oracle as O

dev = torch.device("cuda:0")
c = C.build_case("eval_det_64_128")
n = 32768
ro, rd, bg, _, _ = C.ray_subset(512, 512, 3, n, seed=5)
pc = {k: v.to(dev) for k, v in c["p_coarse"].items()}
pf = {k: v.to(dev) for k, v in c["p_fine"].items()}
ro, rd, bg = ro.to(dev), rd.to(dev), bg.to(dev)
expr, lat = c["expr"].to(dev), c["latent"].to(dev)
_lin = torch.linspace
def run(chunk):
    outs = []
    with torch.no_grad():
        for i in range(0, n, chunk):
            # oracle helpers create a few CPU tensors (linspace); patch by moving inputs: coarse_z builds on CPU -> do it here
            R = min(chunk, n - i)
            z = O.coarse_z(R, O.NEAR, O.FAR, 64, None).to(dev)
            raw = O.paper_mlp(pc, O.encode_points(ro[i:i+R], rd[i:i+R], z, O.NEAR, O.FAR), expr, lat).reshape(R, 64, 4).clone()
            raw[:, -1, :3] = bg[i:i+R]
            rgb_c, _, _, w = O.volume_render(raw, z, rd[i:i+R], None, True)
            zm = 0.5 * (z[:, 1:] + z[:, :-1])
            u = torch.linspace(0, 1, 128, device=dev).expand(R, 128)
            zs = O.sample_pdf(zm, w[:, 1:-1], 128, u)
            zf, _ = torch.sort(torch.cat((z, zs), -1), -1)
            raw = O.paper_mlp(pf, O.encode_points(ro[i:i+R], rd[i:i+R], zf, O.NEAR, O.FAR), expr, lat).reshape(R, 192, 4).clone()
            raw[:, -1, :3] = bg[i:i+R]
            outs.append(O.volume_render(raw, zf, rd[i:i+R], None, True)[0])
    return torch.cat(outs)
for chunk in (2048, 8192):
    run(chunk); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(chunk); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"pytorch-rocm eager fp32, ray chunk {chunk}: {n/dt:,.0f} rays/s ({dt*1e3:.0f} ms for {n} rays)")
