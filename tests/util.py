"""Shared helpers for the GPU parity tests.  TEST INFRASTRUCTURE ONLY."""
import contextlib

import torch

from oracle import nerface_oracle as O


def make_options(nerf, n_coarse, n_fine, perturb, noise_std, chunksize=65536, white=False, lindisp=False):
    mode = dict(num_coarse=n_coarse, num_fine=n_fine, chunksize=chunksize, perturb=perturb, lindisp=lindisp,
                radiance_field_noise_std=noise_std, white_background=white, num_random_rays=2048)
    return nerf.CfgNode(dict(nerf=dict(use_viewdirs=True, encode_position_fn="positional_encoding",
                                       encode_direction_fn="positional_encoding", train=dict(mode), validation=dict(mode)),
                             dataset=dict(no_ndc=True, near=O.NEAR, far=O.FAR)))


def make_model(nerf, params, device):
    m = nerf.models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True,
                                                        include_input_dir=False, use_viewdirs=True, num_layers=4,
                                                        hidden_size=256, include_expression=True)
    m.load_state_dict(params)
    return m.to(device)


def encoders(nerf):
    return (nerf.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True),
            nerf.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True))


@contextlib.contextmanager
def injected_random(rand_list, randn_list):
    """Serve torch.rand / torch.randn from pre-generated tensors, moved to the requested device -- the same
    injection the golden generator applies to the reference (oracle/ref_import.py)."""
    r_it, n_it = iter(rand_list), iter(randn_list)
    keep_r, keep_n = torch.rand, torch.randn

    def serve(it, shape, kw):
        t = next(it)
        want = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert tuple(t.shape) == want, (tuple(t.shape), want)
        return t.clone().to(kw.get("device", "cpu"))

    torch.rand = lambda *s, **kw: serve(r_it, s, kw)
    torch.randn = lambda *s, **kw: serve(n_it, s, kw)
    try:
        yield
    finally:
        torch.rand, torch.randn = keep_r, keep_n


def case_random_lists(c):
    rands, randns = [], []
    if c["stochastic"]:
        rands.append(c["t_rand"])
        if c["noise_std"] > 0:
            randns.append(c["noise_c_unit"])
        if c["n_fine"] > 0:
            rands.append(c["u"])
            if c["noise_std"] > 0:
                randns.append(c["noise_f_unit"])
    return rands, randns


def run_product(nerf, c, device, mode="train", grad=False, chunksize=65536):
    """run_one_iter_of_nerf of the product package on case `c` (see oracle/cases.py)."""
    mc = make_model(nerf, c["p_coarse"], device)
    mf = make_model(nerf, c["p_fine"], device) if c["n_fine"] > 0 else None
    opt = make_options(nerf, c["n_coarse"], c["n_fine"], bool(c["stochastic"]), c["noise_std"], chunksize,
                       lindisp=bool(c.get("lindisp", False)))
    ex, ed = encoders(nerf)
    latent = c["latent"].clone().to(device).requires_grad_(grad)
    rands, randns = case_random_lists(c)
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx, injected_random(rands, randns):
        out = nerf.run_one_iter_of_nerf(512, 512, None, mc, mf, c["ro"].to(device), c["rd"].to(device), opt, mode=mode,
                                        encode_position_fn=ex, encode_direction_fn=ed, expressions=c["expr"].to(device),
                                        background_prior=c["bg"].to(device), latent_code=latent)
    return out, mc, mf, latent


def oracle_render_fp64_on_device(c, ro, rd, bg, device, n_coarse, n_fine, chunk=4096, stages=None, t_rand=None, u=None, mlp=None):
    """The ORACLE (oracle/nerface_oracle.py: torch ops, no product code) evaluated in float64 on `device`, in ray chunks --
    the checker for whole 512x512 frames, which the CPU oracle would need minutes for (mlp: the model family, default the paper model;
    c["p_coarse"] / c["p_fine"] are that family's parameters).  Deterministic sampling unless the
    stratified jitter t_rand (R, n_coarse) and the inverse-CDF abscissae u (R, n_fine) are given (the draws the product is fed)."""
    pc = {k: v.to(device=device, dtype=torch.float64) for k, v in c["p_coarse"].items()}
    pf = {k: v.to(device=device, dtype=torch.float64) for k, v in c["p_fine"].items()}
    expr, lat = c["expr"].to(device=device, dtype=torch.float64), c["latent"].to(device=device, dtype=torch.float64)
    parts = []
    with torch.no_grad():
        for k in range(0, ro.shape[0], chunk):
            f = lambda t: None if t is None else t[k:k + chunk].to(device=device, dtype=torch.float64)
            st = {} if stages is not None else None
            parts.append(O.render_rays(pc, pf, f(ro), f(rd), expr, lat, f(bg), O.NEAR, O.FAR, n_coarse, n_fine, t_rand=f(t_rand), u=f(u),
                                       stages=st, mlp=mlp))
            if stages is not None:
                for name, v in st.items():
                    stages.setdefault(name, []).append(v)
    if stages is not None:
        for name in list(stages):
            stages[name] = torch.cat(stages[name], dim=0)
    return tuple(torch.cat(t, dim=0) for t in zip(*parts))


def parse_bench_stdout(text):
    """bench.py prints two JSON lines: the detail object ({"bench_detail": {...}}) and, LAST, the compact record the driver parses.
    Returns (compact, detail)."""
    import json
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 2, [l[:80] for l in lines]
    detail = json.loads(lines[0])
    assert list(detail) == ["bench_detail"]
    compact = json.loads(lines[1])
    assert len(lines[1]) < 6144 and "bench_detail" not in compact
    return compact, detail["bench_detail"]

