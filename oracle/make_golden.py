"""Generate tests/golden/*.npz by running the UNMODIFIED reference on CPU.  TEST INFRASTRUCTURE ONLY.

Run inside the build container (where /root/reference exists):   python -m oracle.make_golden
The fixtures hold reference OUTPUTS for the seeded synthetic cases of oracle/cases.py; inputs and
weights are regenerated from seeds by the tests (a weight checksum is stored to detect RNG drift).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cases as C                      # noqa: E402
from oracle import nerface_oracle as O             # noqa: E402
from oracle import ref_import as RI                # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def ref_options(ref, n_coarse, n_fine, perturb, noise_std, lindisp=False):
    mode = dict(num_coarse=n_coarse, num_fine=n_fine, chunksize=65536, perturb=perturb, lindisp=lindisp,
                radiance_field_noise_std=noise_std, white_background=False, num_random_rays=2048)
    return ref.CfgNode(dict(nerf=dict(use_viewdirs=True, encode_position_fn="positional_encoding",
                                      encode_direction_fn="positional_encoding", train=dict(mode), validation=dict(mode)),
                            dataset=dict(no_ndc=True, near=O.NEAR, far=O.FAR)))


def ref_model(ref, params):
    m = ref.models.ConditionalBlendshapePaperNeRFModel(
        num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=False,
        use_viewdirs=True, num_layers=4, hidden_size=256, include_expression=True)
    m.load_state_dict(params)
    return m


def run_reference(ref, c, grad=False):
    mc, mf = ref_model(ref, c["p_coarse"]), ref_model(ref, c["p_fine"]) if c["n_fine"] > 0 else None
    opt = ref_options(ref, c["n_coarse"], c["n_fine"], bool(c["stochastic"]), c["noise_std"], bool(c.get("lindisp", False)))
    enc_xyz = ref.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True)
    enc_dir = ref.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True)
    rands, randns = [], []
    if c["stochastic"]:
        rands.append(c["t_rand"])
        if c["noise_std"] > 0:
            randns.append(c["noise_c_unit"])
        if c["n_fine"] > 0:
            rands.append(c["u"])
            if c["noise_std"] > 0:
                randns.append(c["noise_f_unit"])
    latent = c["latent"].clone().requires_grad_(grad)
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx, RI.injected_random(rands, randns), RI.relu_clone_shim(ref):
        out = ref.run_one_iter_of_nerf(512, 512, None, mc, mf, c["ro"], c["rd"], opt, mode="train",
                                       encode_position_fn=enc_xyz, encode_direction_fn=enc_dir,
                                       expressions=c["expr"], background_prior=c["bg"], latent_code=latent)
        grads = None
        if grad:
            loss = O.train_loss(out[0], out[3], c["tgt"], latent)
            loss.backward()
            grads = {"latent": latent.grad.clone(), "loss": loss.detach().clone()}
            for tag, m in (("coarse", mc), ("fine", mf)):
                for k, v in m.named_parameters():
                    grads[f"{tag}.{k}"] = None if v.grad is None else v.grad.clone()
    return out, grads


NOFLIP = "soft_train_noflip_64_64"


def _fp64_autograd(c):
    """The oracle in float64 with autograd on case `c`: (param grads coarse, fine, latent grad, smallest |ReLU input|)."""
    pc = {k: v.double().clone().requires_grad_(True) for k, v in c["p_coarse"].items()}
    pf = {k: v.double().clone().requires_grad_(True) for k, v in c["p_fine"].items()}
    lat = c["latent"].double().clone().requires_grad_(True)
    d = lambda t: None if t is None else t.double()
    keep, rec = torch.relu, []

    def relu(v):                                   # MLP units as they are; the density ReLU (V:52) in units of the x40 head
        rec.append(float(v.detach().abs().min()) / (1.0 if v.shape[-1] in (256, 128) else 40.0))
        return keep(v)
    torch.relu = relu
    try:
        o = O.render_rays(pc, pf, d(c["ro"]), d(c["rd"]), d(c["expr"]), lat, d(c["bg"]), O.NEAR, O.FAR, c["n_coarse"], c["n_fine"],
                          t_rand=d(c["t_rand"]), noise_c=d(c["noise_c"]), u=d(c["u"]), noise_f=d(c["noise_f"]))
    finally:
        torch.relu = keep
    O.train_loss(o[0], o[3], d(c["tgt"]), lat).backward()
    return pc, pf, lat.grad, min(rec)


def _worst_vs_fp64(g, pc, pf):
    rel = lambda a, b: float((a.double() - b).norm() / (b.norm() + 1e-30))
    return max(rel(g[f"{tag}.{k}"], v.grad) for tag, po in (("coarse", pc), ("fine", pf)) for k, v in po.items()
               if g[f"{tag}.{k}"] is not None)


def search_soft_grads(ref, n_frames=2000, keep=8):
    """How oracle/cases.py's NOFLIP frame was chosen.  A ReLU unit whose input lies within rounding of zero takes different sides in two
    fp32 evaluations and moves a gradient tensor by ~1e-3 -- that says nothing about the backward arithmetic.  Rank the frames by their
    smallest |ReLU input| (fp64 oracle, both networks + the density ReLU), then keep those on which the reference's fp32 autograd and the
    oracle's fp64 autograd agree to rounding in EVERY tensor: an fp64 pipeline perturbs every depth and activation at the fp32 rounding
    level, the same size of perturbation another fp32 implementation is allowed, so such a frame has no decision within reach."""
    ranked = []
    for f in range(n_frames):
        C.CASES["_probe"] = dict(C.CASES[NOFLIP], frame=f)
        ranked.append((_fp64_autograd(C.build_case("_probe"))[3], f))
    ranked.sort(reverse=True)
    for margin, f in ranked[:keep]:
        C.CASES["_probe"] = dict(C.CASES[NOFLIP], frame=f)
        c = C.build_case("_probe")
        _, g = run_reference(ref, c, grad=True)
        pc, pf, _, _ = _fp64_autograd(c)
        print(f"frame {f}: smallest |ReLU input| {margin:.2e}, reference fp32 vs oracle fp64 autograd: worst tensor {_worst_vs_fp64(g, pc, pf):.2e}")
    C.CASES.pop("_probe")


def make_soft_grads(ref):
    """soft_train_noflip_64_64_grads.npz: the reference's autograd (Q9 shim) on the soft-family training case, FULL gradient tensors
    of both models and the latent row (rel-L2 <= 1e-4 per tensor, end to end)."""
    c = C.build_case(NOFLIP)
    out, g = run_reference(ref, c, grad=True)
    pc, pf, lat64, margin = _fp64_autograd(c)
    worst = _worst_vs_fp64(g, pc, pf)
    print(f"[{NOFLIP}] loss {float(g['loss']):.6f}; smallest |ReLU input| {margin:.2e}; reference fp32 vs oracle fp64 autograd: worst tensor "
          f"{worst:.2e}, latent {float((g['latent'].double() - lat64).norm() / lat64.norm()):.2e}")
    assert worst < 1e-5, "this frame has a ReLU decision within fp32 rounding: pick another (search_soft_grads)"
    blob = {"loss": g["loss"].numpy(), "latent": g["latent"].numpy(), "relu_margin_fp64": np.float64(margin),
            "params_checksum": np.float64(C.params_checksum(c["p_coarse"]) + C.params_checksum(c["p_fine"]))}
    for n, a in zip(["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_last"], out):
        blob[n] = a.detach().numpy()
    for k, v in g.items():
        if k in ("loss", "latent"):
            continue
        blob[("none:" if v is None else "full:") + k] = np.zeros(0, np.float32) if v is None else v.numpy()
    np.savez_compressed(os.path.join(OUT, f"{NOFLIP}_grads.npz"), **blob)


def lcode_ref_model(ref, params):
    m = ref.models.ConditionalBlendshapeLearnableCodeNeRFModel(
        num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=False, use_viewdirs=True,
        num_layers=4, hidden_size=256, include_expression=True)
    assert list(m.state_dict().keys()) == O.LCODE_KEYS
    m.load_state_dict(params)
    return m


def make_lcode_grads(ref):
    """Training step of the second model family through the reference (autograd, Q9 shim): loss, latent gradient and
    per-tensor gradient norms / heads, same blob format as train_rand_64_64_grads.npz."""
    c = C.build_case("train_rand_64_64")
    pc, pf = O.init_lcode_params(5), O.init_lcode_params(6)
    mc, mf = lcode_ref_model(ref, pc), lcode_ref_model(ref, pf)
    opt = ref_options(ref, c["n_coarse"], c["n_fine"], True, c["noise_std"])
    enc_xyz = ref.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True)
    enc_dir = ref.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True)
    latent = c["latent"].clone().requires_grad_(True)
    with torch.enable_grad(), RI.injected_random([c["t_rand"], c["u"]], [c["noise_c_unit"], c["noise_f_unit"]]), RI.relu_clone_shim(ref):
        out = ref.run_one_iter_of_nerf(512, 512, None, mc, mf, c["ro"], c["rd"], opt, mode="train", encode_position_fn=enc_xyz,
                                       encode_direction_fn=enc_dir, expressions=c["expr"], background_prior=c["bg"],
                                       latent_code=latent)
        loss = O.train_loss(out[0], out[3], c["tgt"], latent)
        loss.backward()
    blob = {"loss": loss.detach().numpy(), "latent": latent.grad.numpy(), "rgb_c": out[0].detach().numpy(), "rgb_f": out[3].detach().numpy()}
    for tag, m in (("coarse", mc), ("fine", mf)):
        for k, v in m.named_parameters():
            g = v.grad
            assert g is not None, k
            blob[f"norm:{tag}.{k}"] = np.float64(g.double().norm())
            blob[f"head:{tag}.{k}"] = g.reshape(-1)[:257].numpy()
    np.savez_compressed(os.path.join(OUT, "lcode_train_rand_64_64_grads.npz"), **blob)
    print("lcode grad fixture: loss", float(loss), "latent |g|", float(latent.grad.norm()))


def eval_post_inputs():
    """Seeded inputs of the eval post-processing fixture: an out-of-range colour image, a disparity-like map and weights
    (some above the 0.22 threshold of EV:112), 48x48 and 64x64."""
    out = {}
    for n, seed in ((48, 8), (64, 9)):
        g = torch.Generator().manual_seed(seed)
        out[n] = (torch.rand((n, n, 3), generator=g) * 1.4 - 0.2, torch.rand((n, n), generator=g) * 0.5 + 1.0,
                  torch.rand((n, n), generator=g) * 0.5)
    return out


def make_eval_post():
    """cast_to_image (EV:184-190) and torch_normal_map(..., clean=True) (EV:84-119) of the UNMODIFIED eval script; the normal
    map is stored as the script hands it to the image writer: `.cpu().numpy().astype('uint8')` (EV:471)."""
    ev = RI.import_reference_eval()
    blob = {}
    for n, (rgb, disp, w) in eval_post_inputs().items():
        blob[f"rgb_u8_{n}"] = ev.cast_to_image(rgb, "blender")
        blob[f"normals_u8_{n}"] = ev.torch_normal_map(disp.clone(), O.INTRINSICS, w, clean=True).cpu().numpy().astype("uint8")
        blob[f"normals_plain_u8_{n}"] = ev.torch_normal_map(disp.clone(), O.INTRINSICS, None, clean=True).cpu().numpy().astype("uint8")
        assert np.array_equal(blob[f"rgb_u8_{n}"], O.cast_to_u8(rgb).numpy())
        assert np.array_equal(blob[f"normals_u8_{n}"], O.normal_map(disp, O.INTRINSICS, w).numpy())
    np.savez_compressed(os.path.join(OUT, "eval_post.npz"), **blob)
    print("eval_post fixture written (oracle restatement == reference, bit exact)")


def make_eval_as_shipped(n_frames=2):
    """The UNMODIFIED eval script run as shipped (EV:201-498: `ablate = 'view_dir'`, pose / expression frozen to test frame 100, view
    directions from pose 240 + i, latent row idx_map[10, 1], background re-read from bg/00050.png) on the CPU of the build container
    against the reference's own `nerf` package, on the synthetic case of run_scripts.as_shipped_case; the uint8 images it hands to
    imageio.imwrite (EV:484-488) are the fixture.  launch/eval_sharded.py --as-shipped must reproduce them."""
    import tempfile
    from oracle import run_scripts as RS
    assert not torch.cuda.is_available(), "the reference script picks cuda when it sees one; the fixture is the CPU result"
    written, blob = {}, {}
    with tempfile.TemporaryDirectory() as tmp, RS.script_stubs(written), RI.flame_loader_io():
        cfg_path, ck_path = RS.as_shipped_case(tmp)
        ev = RS.import_script("eval_transformed_rays", against="reference")
        torch.manual_seed(0)
        RS.run_main(ev, ["--config", cfg_path, "--checkpoint", ck_path, "--savedir", os.path.join(tmp, "out")], max_frames=n_frames)
        for i in range(n_frames):
            blob[f"rgb_u8_{i}"] = written[os.path.join(tmp, "out", f"{i:04d}.png")]
    np.savez_compressed(os.path.join(OUT, "eval_as_shipped.npz"), **blob)
    print("eval_as_shipped fixture:", {k: (v.shape, int(v.sum())) for k, v in blob.items()})


def make_load_flame():
    """load_flame_data (LF:40-211) of the UNMODIFIED reference on the synthetic on-disk dataset of tools/make_synthetic_dataset.py
    (regenerated from its seed by the test), with imageio/cv2 backed as described in ref_import.flame_loader_io."""
    import tempfile
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), "..", "tools"))
    import make_synthetic_dataset as MS
    blob = {}
    with tempfile.TemporaryDirectory() as tmp, RI.flame_loader_io() as lf:
        MS.write(tmp, size=32, n_train=6, n_val=4, n_test=5, seed=3)
        for tag, kw in (("full", dict()), ("half", dict(half_res=True)), ("skip2", dict(testskip=2)), ("test", dict(test=True, half_res=True))):
            imgs, poses, render_poses, hwf, i_split, expr, frontal, bboxs = lf.load_flame_data(tmp, **kw)
            assert frontal is None
            blob.update({f"{tag}_imgs": imgs.numpy(), f"{tag}_poses": poses.numpy(), f"{tag}_render_poses": render_poses.numpy(),
                         f"{tag}_hw": np.array(hwf[:2]), f"{tag}_intrinsics": np.asarray(hwf[2], dtype=np.float64),
                         f"{tag}_expr": expr.numpy(), f"{tag}_bboxs": bboxs.numpy()})
            for k, ix in enumerate(i_split):
                blob[f"{tag}_split{k}"] = np.asarray(ix)
            print(f"load_flame[{tag}]: imgs {tuple(imgs.shape)} {imgs.dtype}, poses {tuple(poses.shape)}, render_poses {render_poses.dtype}, "
                  f"bboxs {bboxs.dtype} {bboxs[0].tolist()}, H W {hwf[:2]}")
    np.savez_compressed(os.path.join(OUT, "load_flame.npz"), **blob)


def make_flex_tiny(ref):
    """BASELINE config 1 read literally ("4-layer MLP"): the UNMODIFIED tiny_nerf.run_one_iter_of_tinynerf (TN:111-159) driving the
    UNMODIFIED nerf.models.FlexibleNeRFModel(num_layers=L, hidden_size=128, num_encoding_fn_xyz=10, use_viewdirs=False) (M:351-422):
    the rendered 64x64 image for L = 2 .. 5 and, for L = 4, one training step (loss + every parameter gradient)."""
    TN = RI.import_reference_tiny()
    pose = O.frame_pose(7)
    pose[:3, 3] = torch.tensor([0.3, -0.2, 4.0])
    focal = torch.tensor(138.88 * 64 / 100.0)
    jit = torch.rand((64, 64, 32), generator=torch.Generator().manual_seed(77))
    target = O.synthetic_image(64, 64, 13)
    blob = {}
    for L in (2, 3, 4, 5):
        fp = O.flex_init_params(4000 + L, L)
        fm = ref.models.FlexibleNeRFModel(num_layers=L, hidden_size=128, num_encoding_fn_xyz=10, include_input_xyz=True, use_viewdirs=False)
        fm.load_state_dict(fp)
        with RI.injected_random([jit], []):
            rgb = TN.run_one_iter_of_tinynerf(64, 64, focal, pose, 2.0, 6.0, 32, lambda x, n: ref.positional_encoding(x, n),
                                              ref.get_minibatches, 16384, fm, 10)
        rgb2, _, _ = O.tiny_render(fp, 64, 64, focal, pose, 2.0, 6.0, 32, 10, jitter=jit)
        print(f"flex tiny L={L}: oracle == reference:", torch.equal(rgb.detach(), rgb2), float((rgb.detach() - rgb2).abs().max()))
        blob[f"rgb_L{L}"] = rgb.detach().numpy()
        if L != 4:
            continue
        loss = torch.nn.functional.mse_loss(rgb, target)
        loss.backward()
        blob["loss_L4"] = loss.detach().numpy()
        for k, v in fm.named_parameters():
            blob["grad_L4:" + k] = v.grad.numpy()
        pp = {k: v.clone().requires_grad_(True) for k, v in fp.items()}
        rgb3, _, _ = O.tiny_render(pp, 64, 64, focal, pose, 2.0, 6.0, 32, 10, jitter=jit)
        torch.nn.functional.mse_loss(rgb3, target).backward()
        worst = max(float((pp[k].grad - torch.from_numpy(blob["grad_L4:" + k])).norm() / (torch.from_numpy(blob["grad_L4:" + k]).norm() + 1e-30))
                    for k in fp)
        print("flex tiny L=4 grad fixture: loss", float(loss), "oracle-vs-reference worst rel L2", worst)
        assert worst < 1e-5
    np.savez_compressed(os.path.join(OUT, "flex_tiny_64x64x32.npz"), **blob)


def make_tiny_grads(ref):
    """One training step of the UNMODIFIED tiny_nerf.py (TN:282-302): rgb = run_one_iter_of_tinynerf(...), loss = mse(rgb, target),
    loss.backward() -- loss and the six parameter gradients (64x64 image, 32 samples, injected jitter)."""
    TN = RI.import_reference_tiny()
    tp = O.tiny_init_params(9458)
    tm = TN.VeryTinyNerfModel(num_encoding_functions=10)
    tm.load_state_dict(tp)
    pose = O.frame_pose(7)
    pose[:3, 3] = torch.tensor([0.3, -0.2, 4.0])
    focal = torch.tensor(138.88 * 64 / 100.0)
    jit = torch.rand((64, 64, 32), generator=torch.Generator().manual_seed(77))
    target = O.synthetic_image(64, 64, 13)
    with RI.injected_random([jit], []):
        rgb = TN.run_one_iter_of_tinynerf(64, 64, focal, pose, 2.0, 6.0, 32, lambda x, n: ref.positional_encoding(x, n),
                                          ref.get_minibatches, 16384, tm, 10)
    loss = torch.nn.functional.mse_loss(rgb, target)
    loss.backward()
    blob = {"loss": loss.detach().numpy(), "rgb": rgb.detach().numpy()}
    for k, v in tm.named_parameters():
        blob["grad:" + k] = v.grad.numpy()
    # the oracle's autograd (fp32) on the same step
    pp = {k: v.clone().requires_grad_(True) for k, v in tp.items()}
    rgb2, _, _ = O.tiny_render(pp, 64, 64, focal, pose, 2.0, 6.0, 32, 10, jitter=jit)
    torch.nn.functional.mse_loss(rgb2, target).backward()
    worst = max(float((pp[k].grad - torch.from_numpy(blob["grad:" + k])).norm() / (torch.from_numpy(blob["grad:" + k]).norm() + 1e-30)) for k in tp)
    print("tiny grad fixture: loss", float(loss), "oracle-vs-reference worst rel L2", worst)
    assert worst < 1e-5
    np.savez_compressed(os.path.join(OUT, "tiny_grads.npz"), **blob)


def make_pe_pdf(ref):
    """positional encoding + sample_pdf_2 (H:344-387) incl. edge cases.  The reference returns only the samples; the CDF
    table and the searchsorted indices stored next to them come from the oracle restatement AFTER it reproduced the
    reference's samples bit for bit on the same inputs (asserted here)."""
    g = torch.Generator().manual_seed(5)
    x = (torch.rand((33, 3), generator=g) - 0.5) * 1.6
    pe10 = ref.positional_encoding(x, 10, True, True)
    pe4 = ref.positional_encoding(x, 4, False, True)
    tu = sys.modules["_ref_nerf.train_utils"]
    bins = torch.sort(torch.rand((6, 63), generator=g) * 0.6 + 0.2, dim=-1)[0]
    w = torch.rand((6, 62), generator=g)
    w[1] = 0.0                      # all-zero weights -> uniform pdf
    w[2, :30] = 0.0
    w[2, 31:] = 0.0                 # single spike
    w[3] = 1.0                      # uniform
    uu = torch.rand((6, 128), generator=g)
    uu[4, :4] = torch.tensor([0.0, 1.0, 0.5, 0.999999])
    with RI.injected_random([uu], []):
        zs_rand = tu.sample_pdf(bins, w, 128, det=False)
    zs_det = tu.sample_pdf(bins, w, 128, det=True)
    tr, td = {}, {}
    assert torch.equal(O.sample_pdf(bins, w, 128, uu, table=tr), zs_rand) and torch.equal(O.sample_pdf(bins, w, 128, None, table=td), zs_det)
    blob = dict(x=x.numpy(), pe10=pe10.numpy(), pe4=pe4.numpy(), bins=bins.numpy(), w=w.numpy(), u=uu.numpy(),
                zs_rand=zs_rand.numpy(), zs_det=zs_det.numpy(), cdf=tr["cdf"].numpy(), inds_rand=tr["inds"].numpy().astype(np.int32),
                inds_det=td["inds"].numpy().astype(np.int32))
    # other table widths: 3 weights (the ragged 5+7 case: torch's scalar row-sum path), 190 (192 coarse samples), 9, 17;
    # compositing-like weights (a few large entries, many exact zeros), 48 rays each
    for nb in (4, 10, 18, 63, 191):
        gg = torch.Generator().manual_seed(100 + nb)
        b = torch.sort(torch.rand((48, nb), generator=gg) * 0.6 + 0.2, dim=-1)[0]
        ww = torch.rand((48, nb - 1), generator=gg) ** 6
        ww[ww < 0.05] = 0.0
        ww = ww / ww.sum(-1, keepdim=True).clamp(min=0.5)
        u2 = torch.rand((48, 64), generator=gg)
        with RI.injected_random([u2], []):
            z2 = tu.sample_pdf(b, ww, 64, det=False)
        t2 = {}
        assert torch.equal(O.sample_pdf(b, ww, 64, u2, table=t2), z2)
        blob.update({f"b{nb}_bins": b.numpy(), f"b{nb}_w": ww.numpy(), f"b{nb}_u": u2.numpy(), f"b{nb}_zs": z2.numpy(),
                     f"b{nb}_inds": t2["inds"].numpy().astype(np.int32), f"b{nb}_cdf": t2["cdf"].numpy()})
    np.savez_compressed(os.path.join(OUT, "pe_pdf.npz"), **blob)
    print("pe_pdf fixture written (oracle == reference samples, bit exact)")


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = RI.import_reference()
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "lcode_grads":        # regenerate only this fixture
        make_lcode_grads(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "soft_grads":         # `soft_grads search`: how the frame of the case was chosen
        search_soft_grads(ref) if sys.argv[2:] == ["search"] else make_soft_grads(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "pe_pdf":
        make_pe_pdf(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "tiny_grads":
        make_tiny_grads(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "flex_tiny":
        make_flex_tiny(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "eval_post":
        make_eval_post()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "load_flame":
        make_load_flame()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "eval_as_shipped":
        make_eval_as_shipped()
        return
    names7 = ["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_last"]
    only = sys.argv[2:] if len(sys.argv) > 2 and sys.argv[1] == "cases" else None     # `cases NAME...`: only these 7-tuple fixtures
    for name in (only or C.CASES):
        c = C.build_case(name)
        out_ref, _ = run_reference(ref, c)
        st = {}
        out_or = C.run_oracle(c, st)
        blob = {"params_checksum": np.float64(C.params_checksum(c["p_coarse"]) + C.params_checksum(c["p_fine"]))}
        for n, a, b in zip(names7, out_ref, out_or):
            if a is None:
                assert b is None
                continue
            exact = torch.equal(a, b)
            print(f"[{name}] {n}: exact={exact} max|d|={float((a - b).abs().max()):.3e}")
            blob[n] = a.numpy()
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **blob)
    if only:
        return

    # ablation path (Quirk Q7, the call pattern of the shipped eval script): 48 rays in 3 chunks of 16, directions for the
    # ENCODING taken from chunk 0 of another pose's rays for every chunk (T:81-82)
    c = C.build_case("eval_det_64_128")
    ro, rd, bg, tgt, idx = C.ray_subset(512, 512, 3, 48, seed=77)
    ro2, rd2 = O.ray_bundle(512, 512, O.INTRINSICS, O.frame_pose(41))
    rd_abl = rd2.reshape(-1, 3)[idx].contiguous()
    c.update(n_rays=48, ro=ro, rd=rd, bg=bg, tgt=tgt, idx=idx)
    mc, mf = ref_model(ref, c["p_coarse"]), ref_model(ref, c["p_fine"])
    opt = ref_options(ref, 64, 128, False, 0.0)
    opt.nerf.train.chunksize = 16
    enc_xyz = ref.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True)
    enc_dir = ref.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True)
    with torch.no_grad():
        out_ref = ref.run_one_iter_of_nerf(512, 512, None, mc, mf, ro.clone(), rd.clone(), opt, mode="train", encode_position_fn=enc_xyz,
                                           encode_direction_fn=enc_dir, expressions=c["expr"], background_prior=bg,
                                           latent_code=c["latent"], ray_directions_ablation=rd_abl.clone())
    parts = [O.render_rays(c["p_coarse"], c["p_fine"], ro[k:k + 16], rd[k:k + 16], c["expr"], c["latent"], bg[k:k + 16], O.NEAR, O.FAR,
                           64, 128, rd_view=rd_abl[:16]) for k in range(0, 48, 16)]          # per ray chunk, like the reference
    out_or = [torch.cat(t, dim=0) for t in zip(*parts)]
    # (the reference also feeds the MLP 16 POINTS at a time here -- chunksize is one number, T:20 -- so GEMM blocking differs
    # from the oracle's and the comparison is to fp32 rounding, not bit-exact)
    print("ablation max|d| coarse rgb:", float((out_ref[0] - out_or[0]).abs().max()), "fine rgb:", float((out_ref[3] - out_or[3]).abs().max()))
    assert float((out_ref[0] - out_or[0]).abs().max()) < 1e-6
    np.savez_compressed(os.path.join(OUT, "ablation_64_128.npz"), **{n: a.numpy() for n, a in zip(names7, out_ref)})

    # second model family (ConditionalBlendshapeLearnableCodeNeRFModel), eval forward, built as the trainer builds it
    c = C.build_case("eval_det_64_128")
    pc, pf = O.init_lcode_params(5), O.init_lcode_params(6)
    def lmodel(params):
        m = ref.models.ConditionalBlendshapeLearnableCodeNeRFModel(
            num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=False, use_viewdirs=True,
            num_layers=4, hidden_size=256, include_expression=True)
        assert list(m.state_dict().keys()) == O.LCODE_KEYS
        m.load_state_dict(params)
        return m
    opt = ref_options(ref, 64, 128, False, 0.0)
    with torch.no_grad():
        out_ref = ref.run_one_iter_of_nerf(512, 512, None, lmodel(pc), lmodel(pf), c["ro"], c["rd"], opt, mode="train",
                                           encode_position_fn=enc_xyz, encode_direction_fn=enc_dir, expressions=c["expr"],
                                           background_prior=c["bg"], latent_code=c["latent"])
    out_or = O.render_rays(pc, pf, c["ro"], c["rd"], c["expr"], c["latent"], c["bg"], O.NEAR, O.FAR, 64, 128, mlp=O.lcode_mlp)
    print("lcode exact:", all(torch.equal(a, b) for a, b in zip(out_ref, out_or)), "w_last range", float(out_ref[6].min()), float(out_ref[6].max()))
    np.savez_compressed(os.path.join(OUT, "lcode_eval_det_64_128.npz"), **{n: a.numpy() for n, a in zip(names7, out_ref)})
    make_lcode_grads(ref)
    make_soft_grads(ref)

    # gradient fixture (reference autograd with the Q9 shim)
    c = C.build_case("train_rand_64_64")
    out_ref, g = run_reference(ref, c, grad=True)
    blob = {"loss": g["loss"].numpy(), "latent": g["latent"].numpy()}
    for k, v in g.items():
        if k in ("loss", "latent"):
            continue
        if v is None:
            blob["none:" + k] = np.zeros(0, np.float32)
        else:
            blob["norm:" + k] = np.float64(v.double().norm())
            if v.numel() <= 1024:
                blob["full:" + k] = v.numpy()
            else:
                blob["head:" + k] = v.reshape(-1)[:257].numpy()
    np.savez_compressed(os.path.join(OUT, "train_rand_64_64_grads.npz"), **blob)
    print("grad fixture: loss", float(g["loss"]), "latent |g|", float(g["latent"].norm()))

    # ray bundle (non-square, full) + bit-exactness of the oracle restatement
    pose = O.frame_pose(42)
    ro, rd = ref.get_ray_bundle(37, 53, O.INTRINSICS, pose)
    ro2, rd2 = O.ray_bundle(37, 53, O.INTRINSICS, pose)
    print("ray_bundle exact:", torch.equal(rd, rd2), torch.equal(ro, ro2))
    ro_s, rd_s = ref.get_ray_bundle(24, 24, torch.tensor(138.88 * 24 / 100.0), pose)     # scalar-focal fallback (H:109)
    np.savez_compressed(os.path.join(OUT, "ray_bundle.npz"), rd=rd.numpy(), ro=ro.numpy(), rd_scalar=rd_s.numpy())

    make_pe_pdf(ref)
    make_eval_post()
    make_load_flame()
    make_tiny_grads(ref)
    make_flex_tiny(ref)

    # tiny_nerf (BASELINE config 1): 64x64, 32 samples, 3-layer 128-wide MLP, coarse only
    TN = RI.import_reference_tiny()
    tp = O.tiny_init_params(9458)
    tm = TN.VeryTinyNerfModel(num_encoding_functions=10)
    tm.load_state_dict(tp)
    pose = O.frame_pose(7)
    pose[:3, 3] = torch.tensor([0.3, -0.2, 4.0])
    focal = torch.tensor(138.88 * 64 / 100.0)
    jit = torch.rand((64, 64, 32), generator=torch.Generator().manual_seed(77))
    with torch.no_grad(), RI.injected_random([jit], []):
        rgb = TN.run_one_iter_of_tinynerf(64, 64, focal, pose, 2.0, 6.0, 32,
                                          lambda x, n: ref.positional_encoding(x, n), ref.get_minibatches, 16384, tm, 10)
    rgb2, _, _ = O.tiny_render(tp, 64, 64, focal, pose, 2.0, 6.0, 32, 10, jitter=jit)
    print("tiny exact:", torch.equal(rgb, rgb2), float((rgb - rgb2).abs().max()))
    np.savez_compressed(os.path.join(OUT, "tiny_64x64x32.npz"), rgb=rgb.numpy())


if __name__ == "__main__":
    main()
