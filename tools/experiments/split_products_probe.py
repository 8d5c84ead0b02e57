"""which of the three products of the split arithmetics (W_lo x_hi, W_hi x_lo, W_hi x_hi) can be dropped on which layers?
Variants of the split INFERENCE kernels are built as lib/libnerface_hip_<name>.so with -DNFB_PRODUCT_MASKS=... (3 bits per layer, layer 0 in
the low bits; bit 0 = W_lo x_hi, bit 1 = W_hi x_lo, bit 2 = W_hi x_hi; the fc_alpha tile of layers_dir.0 always keeps all three).
    python tools/split_products_probe.py build           (CPU: all variants)
    python tools/split_products_probe.py run             (GPU: one subprocess per variant, markdown table on stdout)
    python tools/split_products_probe.py one NAME        (GPU: this process, library chosen by NERFACE_HIP_LIB)
Per variant and arithmetic: fine-launch time (65536 x 192), rms error of the MLP outputs against the exact-f32 kernel on the same inputs,
whole-frame |dPSNR| against a random target and self-PSNR against the exact-f32 frame (same draws), sustained clock / busy cycles (PMC)."""
import json, math, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
ALL, L = 0x1FFFFFFFF, lambda layers, m: sum(((m ^ 7) << (3 * l)) for l in layers)      # masks: all-ones minus the dropped bits
DIR, XYZ = (7, 8, 9, 10), (0, 1, 2, 3, 4, 5, 6)
mk = lambda layers, m: ALL ^ L(layers, m)
VARIANTS = [("base", ALL), ("dir_no_Wlo", mk(DIR, 6)), ("dir_no_xlo", mk(DIR, 5)), ("dir_main_only", mk(DIR, 4)),
            ("h5_no_Wlo", mk((5,), 6)), ("feat_no_Wlo", mk((6,), 6)), ("all_no_Wlo", mk(DIR + XYZ, 6)), ("all_no_xlo", mk(DIR + XYZ, 5)),
            ("all_main_only", mk(DIR + XYZ, 4))]


def lib_of(name):
    return os.path.join(R, "4d-facial-avatars_amd", "lib", f"libnerface_hip_x{name}.so")


def one(name):
    import torch, bench, nerf
    from nerf import ops
    dev = torch.device("cuda:0")
    out = {"name": name}
    mc, mf = bench.synth_params(0, dev), bench.synth_params(1, dev)
    # (1) MLP level: 65536 x 192 launch, time + error against the exact-f32 kernel
    ro, rd = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(0).to(dev))
    n_rays, S = 65536, 192
    ro_, rd_ = ro.view(-1, 3)[:n_rays].contiguous(), rd.view(-1, 3)[:n_rays].contiguous()
    g = torch.Generator().manual_seed(1000)
    expr, lat = (0.5 * torch.randn(76, generator=g)).to(dev), (0.1 * torch.randn(32, generator=g)).to(dev)
    z = torch.sort(torch.rand((n_rays, S), device=dev) * 0.6 + 0.2, dim=-1)[0].contiguous()
    hw = mf.hip_weights()
    cond = ops.paper_condition(hw.get(), expr, lat, 0.2, 0.8)
    ref = ops.paper_mlp_fwd(hw.get(), cond, ro_, rd_, z).double()
    scale = ref.reshape(-1, 4).abs().mean(0)
    for prec, fn in (("f16x3", lambda: ops.paper_mlp_fwd_f16(hw.get_f16(), cond, ro_, rd_, z)),
                     ("bf16x3", lambda: ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro_, rd_, z))):
        raw = fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): fn()
        e1.record(); torch.cuda.synchronize()
        rms = ((raw.double() - ref) ** 2).reshape(-1, 4).mean(0).sqrt()
        out[prec] = {"launch_ms": e0.elapsed_time(e1) / 8, "rms_vs_f32_kernel": [float(v) for v in rms], "mean_abs_output": [float(v) for v in scale]}
    # (2) whole frame through the product pipeline: same draws (seeded) in every arithmetic
    opt = bench.options(nerf)
    ex = nerf.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True)
    ed = nerf.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True)
    bg = torch.rand((512, 512, 3), generator=torch.Generator().manual_seed(7)).to(dev).view(-1, 3)
    tgt = torch.rand((512, 512, 3), generator=torch.Generator().manual_seed(11)).to(dev)
    frames = {}
    for prec in ("f32", "f16x3", "bf16x3"):
        nerf.set_mlp_precision(prec)
        torch.manual_seed(4321)
        with torch.no_grad():
            o = nerf.run_one_iter_of_nerf(512, 512, bench.INTRINSICS, mc, mf, ro, rd, opt, mode="validation", encode_position_fn=ex,
                                          encode_direction_fn=ed, expressions=expr, background_prior=bg, latent_code=lat)
        frames[prec] = o[3].double()
    nerf.set_mlp_precision("f32")
    psnr = lambda a, b: -10.0 * math.log10(float(((a - b) ** 2).mean()))
    for prec in ("f16x3", "bf16x3"):
        out[prec]["abs_dpsnr_db"] = abs(psnr(frames[prec], tgt.double()) - psnr(frames["f32"], tgt.double()))
        out[prec]["self_psnr_db"] = psnr(frames[prec], frames["f32"])
        out[prec]["max_abs_rgb_diff"] = float((frames[prec] - frames["f32"]).abs().max())
    # (3) the clock the f16x3 kernel held and its busy cycles (one PMC pass; inherits NERFACE_HIP_LIB)
    mhz, det = bench.pmc_sustained_clock("f16x3")
    out["f16x3"]["sustained_clock_mhz"] = mhz
    if isinstance(det, dict) and det.get("busy_cycles_raw_median"):
        out["f16x3"]["busy_mcycles"] = det["busy_cycles_raw_median"] / det["xcd_sum_divisor"] / 1e6
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "run"
    if cmd == "build":
        import build
        for name, masks in VARIANTS:
            build.build_variant("x" + name, [f"-DNFB_PRODUCT_MASKS=0x{masks:X}ull"], verbose=False)
            print("built", name, hex(masks), flush=True)
    elif cmd == "one":
        one(sys.argv[2])
    else:
        rows = []
        for name, masks in VARIANTS:
            if not os.path.exists(lib_of(name)):
                print("missing", lib_of(name)); continue
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", name], env=dict(os.environ, NERFACE_HIP_LIB=lib_of(name)),
                               capture_output=True, text=True, timeout=600)
            hit = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not hit:
                print(name, "FAILED", r.stderr[-600:]); continue
            rows.append((masks, json.loads(hit[0][7:])))
        n_mfma = lambda m: sum(bin((m >> (3 * l)) & 7).count("1") * k for l, k in enumerate((32, 128, 128, 160, 128, 128, 128, 80, 32, 32, 8))) + 60
        print("| variant | masks | MFMAs / 32 pts | f16x3 ms | clock MHz | busy Mcycles | f16x3 rms rgb / sigma vs f32 kernel | f16x3 abs dPSNR dB | self-PSNR dB | bf16x3 ms | bf16x3 abs dPSNR dB | self-PSNR dB |")
        print("|---|---|---|---|---|---|---|---|---|---|---|---|")
        for masks, o in rows:
            f, b = o["f16x3"], o["bf16x3"]
            print(f"| {o['name']} | 0x{masks:X} | {n_mfma(masks)} | {f['launch_ms']:.2f} | {f.get('sustained_clock_mhz') or 0:.0f} | {f.get('busy_mcycles') or 0:.1f} | "
                  f"{max(f['rms_vs_f32_kernel'][:3]):.2e} / {f['rms_vs_f32_kernel'][3]:.2e} | {f['abs_dpsnr_db']:.2e} | {f['self_psnr_db']:.1f} | "
                  f"{b['launch_ms']:.2f} | {b['abs_dpsnr_db']:.2e} | {b['self_psnr_db']:.1f} |")
        print(json.dumps([o for _, o in rows]))
