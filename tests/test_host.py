"""CPU: the C-ABI library loads and exports every symbol include/nerface_hip.h declares; host-side logic."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "nerface_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nf_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 17
    for name in sorted(declared):
        assert hasattr(hip_lib, name), f"libnerface_hip.so does not export {name}"
    from nerf import _hip
    assert hip_lib.nf_abi_version() == _hip.ABI_VERSION == 5          # exact: the ctypes prototypes are written for ONE revision
    assert b"gfx950" in hip_lib.nf_build_info()
    assert hip_lib.nf_error_string(-22).startswith(b"nerface_hip")


def test_gather_table_covers_every_live_weight_once(hip_lib):
    """Every live weight element must appear exactly once among the MFMA fragment sections (the conditioning
    matrices and the bias table re-reference some), and nothing may reference the dead layers_dir.3."""
    from nerf import ops
    n = hip_lib.nf_paper_packed_floats()
    tab = np.zeros(n, dtype=np.uint32)
    assert hip_lib.nf_paper_gather_table(tab.ctypes.data_as(ctypes.c_void_p), n) == 0
    ids, offs = tab >> 24, tab & 0xFFFFFF
    import nerf
    m = nerf.models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True,
                                                        include_input_dir=False)
    shapes = [tuple(p.shape) for p in m.hip_param_list()]
    assert len(shapes) == 26 and list(dict(m.named_parameters())) == ops.PAPER_KEYS
    frag_end = 499968                                   # nfl::FRAG_END
    assert not np.any((ids == 22) | (ids == 23))        # layers_dir.3.{weight,bias} (Quirk Q3)
    for tid, shp in enumerate(shapes):
        if tid in (22, 23):
            continue
        numel = int(np.prod(shp))
        sel = offs[(ids == tid)]
        assert sel.max() < numel
        if len(shp) == 2:                                # weights: MFMA sections cover the non-folded columns once
            cnt = np.bincount(offs[:frag_end][ids[:frag_end] == tid], minlength=numel).reshape(shp)
            if tid == 0:                                 # layers_xyz.0: cols 0..62 in fragments, 63..170 folded
                assert np.all(cnt[:, :63] == 1) and np.all(cnt[:, 63:] == 0)
            elif tid == 6:                               # layers_xyz.3: [pe 63 | cond 108 | h 256]
                assert np.all(cnt[:, :63] == 1) and np.all(cnt[:, 63:171] == 0) and np.all(cnt[:, 171:] == 1)
            elif tid == 16:                              # layers_dir.0: feat + the 8 rd_z columns in fragments
                varying = [256 + 6 * f + 3 * sc for f in range(4) for sc in range(2)]
                assert np.all(cnt[:, :256] == 1) and np.all(cnt[:, varying] == 1)
                const = [c for c in range(256, 280) if c not in varying]
                assert np.all(cnt[:, const] == 0)
            else:
                assert np.all(cnt == 1), tid
        else:
            cnt = np.bincount(offs[frag_end:][ids[frag_end:] == tid], minlength=numel)
            assert np.all(cnt == 1), tid
    # the folded columns are present exactly once in the conditioning matrices
    cnt0 = np.bincount(offs[frag_end:][ids[frag_end:] == 0], minlength=256 * 171).reshape(256, 171)
    assert np.all(cnt0[:, 63:] == 1) and np.all(cnt0[:, :63] == 0)


def test_weight_gradient_job_tables_are_consistent(hip_lib):
    """Host-only self-tests of the dW job tables (exact-f32 wave jobs and split-bf16 workgroup bundles, both model families):
    every slab entry the unpack kernels read is written exactly once per slice, nothing twice, tile ids inside their bundle."""
    assert hip_lib.nf_selftest_dw_tables_f32() == 0
    assert hip_lib.nf_selftest_dw_tables_lcode_f32() == 0
    assert hip_lib.nf_selftest_dw_tables_bf16() == 0


def _stream_table(hip_lib, fn):
    n = getattr(hip_lib, fn)(None, 0)
    assert n > 0
    tab = np.zeros(n, dtype=np.uint32)
    assert getattr(hip_lib, fn)(tab.ctypes.data_as(ctypes.c_void_p), n) == n
    return tab >> 24, tab & 0xFFFFFF


def _check_stream(ids, offs, shapes, expect):
    """expect: tensor id -> boolean array (tensor shape): which elements must appear exactly once; all others never."""
    for tid, shp in enumerate(shapes):
        cnt = np.bincount(offs[ids == tid], minlength=int(np.prod(shp))).reshape(shp)
        want = expect.get(tid)
        if want is None:
            assert cnt.sum() == 0, tid
        else:
            assert np.array_equal(cnt, want.astype(cnt.dtype)), (tid, int((cnt != want).sum()))
    assert set(np.unique(ids)) <= set(expect) | {0xFF}


def test_split_bf16_weight_streams_cover_the_right_elements_once(hip_lib):
    """Gather tables of the four split-bf16 weight streams: every weight element the kernels multiply by appears exactly once,
    the folded columns (expression, latent, PE of near/far: they live in the per-call bias table) and the biases never, the
    dead layers_dir.3 never; the transposed streams hold exactly the weights the dX chain needs."""
    import nerf
    full = lambda shp: np.ones(shp, dtype=bool)
    cols = lambda shp, sel: np.broadcast_to(np.isin(np.arange(shp[1]), list(sel)), shp)
    dir_live = list(range(256)) + [256 + 6 * f + 3 * sc for f in range(4) for sc in range(2)]       # feat + sin/cos of rd_z
    # ---- paper model (ids = position in ops.PAPER_KEYS: weight 2 i, bias 2 i + 1)
    m = nerf.models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True,
                                                        include_input_dir=False)
    shapes = [tuple(p.shape) if p.dim() == 2 else (1, p.numel()) for p in m.hip_param_list()]
    fwd = {0: cols(shapes[0], range(63)), 2: full(shapes[2]), 4: full(shapes[4]),
           6: cols(shapes[6], list(range(63)) + list(range(171, 427))), 8: full(shapes[8]), 10: full(shapes[10]), 12: full(shapes[12]),
           14: full(shapes[14]), 16: cols(shapes[16], dir_live), 18: full(shapes[18]), 20: full(shapes[20]), 24: full(shapes[24])}
    _check_stream(*_stream_table(hip_lib, "nf_paper_stream_table_bf16"), shapes, fwd)
    bwd = {24: full(shapes[24]), 20: full(shapes[20]), 18: full(shapes[18]), 16: cols(shapes[16], range(256)), 14: full(shapes[14]),
           12: full(shapes[12]), 10: full(shapes[10]), 8: full(shapes[8]), 6: cols(shapes[6], range(171, 427)), 4: full(shapes[4]),
           2: full(shapes[2])}                                          # layers_xyz.0 feeds no dX
    _check_stream(*_stream_table(hip_lib, "nf_paper_stream_table_bwd_bf16"), shapes, bwd)
    # ---- second family (ids = position in models.LCODE_KEYS)
    ml = nerf.models.ConditionalBlendshapeLearnableCodeNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True,
                                                                 include_input_dir=False, num_layers=4, hidden_size=256)
    shp = [tuple(p.shape) if p.dim() == 2 else (1, p.numel()) for p in ml.hip_param_list()]
    fwd = {0: cols(shp[0], range(63)), 2: full(shp[2]), 4: full(shp[4]), 6: full(shp[6]), 8: cols(shp[8], dir_live), 10: full(shp[10]),
           12: full(shp[12]), 14: full(shp[14])}
    _check_stream(*_stream_table(hip_lib, "nf_lcode_stream_table_bf16"), shp, fwd)
    bwd = {12: full(shp[12]), 8: cols(shp[8], range(256)), 14: full(shp[14]), 10: full(shp[10]), 6: full(shp[6]), 4: full(shp[4]),
           2: full(shp[2])}                                             # layer1 feeds no dX
    _check_stream(*_stream_table(hip_lib, "nf_lcode_stream_table_bwd_bf16"), shp, bwd)


def test_state_dict_schema_matches_reference_checkpoints():
    import nerf
    m = nerf.models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True,
                                                        include_input_dir=False, use_viewdirs=True, num_layers=4,
                                                        hidden_size=256, include_expression=True)
    from oracle import nerface_oracle as O
    sd = m.state_dict()
    assert list(sd.keys()) == O.PAPER_KEYS
    for k, shp in O.PAPER_SHAPES.items():
        assert tuple(sd[k].shape) == shp and tuple(sd[k.replace("weight", "bias")].shape) == (shp[0],)
    assert sum(v.numel() for v in sd.values()) == 568708
    m.load_state_dict(O.init_paper_params(3))          # reference-style checkpoint dict round-trips
    assert m.fused_supported()


def test_product_has_no_cpu_path():
    """CPU tensors must raise (no silent fallback), for the kernels and for the model's own forward."""
    import nerf
    from oracle import nerface_oracle as O
    with pytest.raises(RuntimeError):
        nerf.get_ray_bundle(4, 4, O.INTRINSICS, O.frame_pose(0))
    with pytest.raises(RuntimeError):
        nerf.positional_encoding(torch.zeros(3, 3), 10, True)
    m = nerf.models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_dir=False)
    with pytest.raises(NotImplementedError):
        m(torch.zeros(2, 87), torch.zeros(76), torch.zeros(32))               # grad mode: no autograd through forward()
    with torch.no_grad(), pytest.raises(RuntimeError):
        m(torch.zeros(2, 87), torch.zeros(76), torch.zeros(32))               # CPU tensors: no CPU path


def test_oracle_is_imported_only_by_the_checkers():
    """oracle/ is test infrastructure: nothing in the product package, the launchers, tools/, bench.py, bench_common.py or bench_probes.py
    may import it; bench_baselines.py -- the baseline legs of the bench line -- may, and only inside the functions that ARE baselines
    (cpu_baseline + its helper _reference_cpu_run, cpu_baseline_tiny, eager_rocm_baseline, eager_rocm_reference);
    __graft_entry__ only as the smoke() / build() checker."""
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)", re.M)
    for base in ("4d-facial-avatars_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dirpath, f)).read()
                    assert not pat.search(src), os.path.join(dirpath, f)
    for f in ("bench.py", "bench_common.py", "bench_probes.py"):
        assert not pat.search(open(os.path.join(ROOT, f)).read()), f
    legs_src = open(os.path.join(ROOT, "bench_baselines.py")).read()
    n_legs = 0
    for fn in ("def _reference_cpu_run(", "def cpu_baseline(", "def cpu_baseline_tiny(", "def eager_rocm_baseline(", "def eager_rocm_reference("):
        leg = legs_src[legs_src.index(fn):]
        leg = leg[:leg.index("\ndef ", 1)] if "\ndef " in leg[1:] else leg
        n_legs += len(pat.findall(leg))
    assert len(pat.findall(legs_src)) == n_legs > 0


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("nf_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_compact_line_fits_the_driver_record():
    """the final stdout line of bench.py must be a record the driver can keep whole (its tail buffer is ~8 KB; the
    round-4 line was 24 KB and came back `parsed: null`).  Built here from a canned full result (the round-4 line, every new
    summary key filled with a worst-case 17-digit float): < 6144 bytes, the contract's keys, flat `roofline` / `cpu_baseline`."""
    import json
    B = _load_bench()
    line = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    line["launcher"] = {"launcher_eval_frames_s": 2.1234567890123457, "launcher_gpu_s_per_frame": 0.46123456789012345, "launcher_wall_over_gpu": 1.0123456789012345}
    line["config"]["device"]["pattern_store"] = {"stream_nt_gbs": 4321.123456789012, "rows_nt_gbs": 4321.123456789012, "seq_nt_gbs": 4321.123456789012,
                                                  "first_touch_stream_nt_gbs": 321.1234567890123, "stream_gbs": 4321.123456789012,
                                                  "rows_gbs": 4321.123456789012}
    kern = {"power_w": 1334.1234567890123, "sclk_mhz_hwmon": 2204.1234567890123, "fclk_mhz_dpm": 1250.1234567890123}
    line["config"]["device"]["power"] = {"static": {"power_cap_w": 1400.1234567890123, "perf_level": "auto"}, "f32": kern, "f16x3": kern, "f16x2": kern, "bf16x3": kern, "train_fwd_bf16x3": dict(kern, launch_ms=1.1234567890123457),
                                         "train_fwd_f32": dict(kern, launch_ms=2.1234567890123457)}
    line["eager_rocm"].update(kind="reference", port={"value": 255026.83364130167})
    import copy
    line["split_f16x2"] = copy.deepcopy(line["split_f16"])                 # round 5: the fourth arithmetic
    cells = {t: {m: 3.5123456789012345e-05 for m in ("whole", "3001", "1024")} for t in ("random", "20dB", "30dB", "40dB")}
    for k in ("split_f16x2", "split_f16", "split_bf16"):                  # round 6: the gate per (target, ray count), nerf/gate.py
        line[k]["gate_vs_exact_f32"] = {"frames_checked": 4, "min_self_psnr_db": 67.41234567890123, "max_self_psnr_db": 77.41234567890123,
                                        "worst_abs_dpsnr_db": cells}
        line[k]["roofline"].setdefault("frac_executed", 0.49123456789012345)
        line[k]["roofline"].setdefault("sustained_clock_mhz", 2134.1234567890123)
    line["per_rank_ms_per_step"] = [455.12345678901234]
    line["eager_rocm"].update(value_min=231234.56789012345, value_max=241234.56789012345, frames=3)
    for prec in ("f16x3", "bf16x3"):                                      # round 6: executed 16-bit MFMA fraction of the split training kernels
        for k in line["train"][prec]["roofline"]["kernels"]:
            k["frac_executed_mfma"] = 0.31234567890123456
    line["cpu_baseline"]["parity_on_sample"]["f16x2"] = dict(line["cpu_baseline"]["parity_on_sample"]["f16x3"])
    line["tiny"]["cpu_baseline"].update(kind="reference")
    line["tiny4"] = copy.deepcopy(line["tiny"])                            # round 6: configs[0] read literally (4-layer FlexibleNeRFModel)
    line["summary"] = B.summary_of(line)
    assert all(v is not None for k, v in line["summary"].items() if k != "train_allreduce_us"), [k for k, v in line["summary"].items() if v is None]
    c = B.compact_line(line)
    text = json.dumps(c)
    assert len(text) < B.COMPACT_LIMIT == 6144, len(text)
    assert "summary_truncated" not in c
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "summary", "ranks_seen"):
        assert k in c, k
    assert list(c["config"]) == ["workload"] and list(c)[-1] == "summary"
    assert set(c["roofline"]) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "frac_algorithmic", "avg_launch_ms", "traffic",
                                  "algorithmic_hbm_bytes_per_launch", "sustained_clock_mhz"}
    assert set(c["cpu_baseline"]) == {"value", "unit", "kind", "cores", "host_cores", "sample"} and c["cpu_baseline"]["kind"] == "reference"
    assert all(not isinstance(v, (dict, list)) for v in c["summary"].values())
    assert all(not isinstance(v, (dict, list)) for k, v in c["roofline"].items())
    assert abs(c["value"] - line["value"]) <= 1e-5 * line["value"] and c["roofline"]["frac"] == round(line["roofline"]["frac"], 6)
    for k in ("train_ms_bf", "tr_bf_fwd_ms_2400_est", "tr_bf_fwd_frac_mfma", "tr_f16_chain_frac_mfma", "f16_rays_s", "pattern_store_gbs", "product_over_eager",
              "launcher_eval_frames_s", "eager_rocm_kind", "eager_rocm_min_rays_s", "tiny_cpu_kind", "tiny4_rays_s", "tiny4_cpu_rays_s", "power_cap_w", "w_f16", "mhz_f16", "x2_rays_s",
              "matched_psnr_over_eager_f16x3", "speed_only_over_eager_f16x2", "f16_gate_30db_worst_db", "x2_gate_30db_worst_db", "bf_gate_30db_worst_db",
              "f16_gate_30db_1024rays_db", "x2_gate_random_db", "f16_self_psnr_min_db", "rank_ms_max", "cpu_threads"):
        assert k in c["summary"], k
    assert "split_f16x2_over_eager" not in c["summary"]                   # (round 5's name read as a matched-PSNR ratio)
    # an overgrown summary sheds keys from the back instead of breaking the record
    line["summary"].update({f"pad_{i}": "x" * 40 for i in range(200)})
    c2 = B.compact_line(line)
    assert len(json.dumps(c2)) <= B.COMPACT_LIMIT and c2["summary_truncated"] and "value_rays_s" in c2["summary"]


def test_bench_gpus_flag_is_honoured():
    """`--gpus N` without a launcher re-executes under torch.distributed.run with N ranks on 127.0.0.1; a launcher whose
    WORLD_SIZE disagrees with --gpus is refused before anything runs (no GPU needed for either check)."""
    B = _load_bench()
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if "WORLD_SIZE" not in os.environ and args.gpus > 1' in src and "self_launch(args.gpus)" in src
    calls = []
    keep = B.subprocess.call
    B.subprocess.call = lambda cmd, *a, **k: calls.append(cmd) or 0
    try:
        assert B.self_launch(4) == 0
    finally:
        B.subprocess.call = keep
    cmd = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(os.environ, WORLD_SIZE="1", RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"must agree" in r.stderr


def test_tiny_dw_job_table_selftest(hip_lib):
    """Weight-gradient job table of the tiny path (three products): every slab entry written exactly once."""
    assert hip_lib.nf_selftest_dw_tables_tiny() == 0


def test_flex_tiny_host_side(hip_lib):
    """The literal "4-layer MLP" of BASELINE config 1 (FlexibleNeRFModel behind the tiny path, nf_flex_*), host side only: the job table
    of every supported depth writes each slab entry exactly once; sizes follow the layer count; unsupported depths are refused; the
    product's model class has the reference's parameter names / shapes (M:351-394) and knows which geometries have kernels."""
    import nerf
    for L in (2, 3, 4, 5):
        assert hip_lib.nf_selftest_dw_tables_flex(L) == 0
        nh = L - 1
        assert hip_lib.nf_flex_grad_floats(L) == 128 * 63 + 128 + nh * (128 * 128 + 128) + 4 * 128 + 4
        assert hip_lib.nf_flex_saved_floats(L, 10) == 10 * (64 + 128 * L)
        assert hip_lib.nf_flex_packed_floats(L) == 8192 + nh * 16384 + 2048 + 128 * L + 16
        assert hip_lib.nf_flex_packed_bwd_floats(L) == 2048 + nh * 16384
        assert hip_lib.nf_flex_bwd_workspace_floats(L, 4096 * 32) > 128 * L * 4096 * 32
        m = nerf.models.FlexibleNeRFModel(num_layers=L, hidden_size=128, num_encoding_fn_xyz=10, include_input_xyz=True, use_viewdirs=False)
        assert m.fused_supported() and m.num_layers == L
        assert sum(p.numel() for p in m.hip_param_list()) == hip_lib.nf_flex_grad_floats(L)
        assert [k for k, _ in m.named_parameters()] == (["layer1.weight", "layer1.bias"] + [f"layers_xyz.{i}.{w}" for i in range(nh) for w in ("weight", "bias")]
                                                        + ["fc_out.weight", "fc_out.bias"])
    for L in (0, 1, 6, 8):
        assert hip_lib.nf_flex_packed_floats(L) == 0 and hip_lib.nf_flex_grad_floats(L) == 0 and hip_lib.nf_selftest_dw_tables_flex(L) == -22
        assert hip_lib.nf_flex_mlp_fwd(L, None, None, None, None, 1, 4, 4, None, None) == -22
    assert hip_lib.nf_flex_mlp_fwd(4, None, None, None, None, 1, 0, 4, None, None) == 0                      # no rays: nothing to do
    assert hip_lib.nf_flex_mlp_fwd(4, None, None, None, None, 1, 4, 4, None, None) == -22
    # geometries without a kernel construct (state_dict compatibility, M:351-394) and say so
    full = nerf.models.FlexibleNeRFModel()                                                            # the reference's defaults: view directions
    assert not full.fused_supported() and set(dict(full.named_parameters())) >= {"layers_dir.0.weight", "fc_alpha.weight", "fc_rgb.weight", "fc_feat.weight"}
    deep = nerf.models.FlexibleNeRFModel(num_layers=8, hidden_size=128, num_encoding_fn_xyz=10, use_viewdirs=False)
    assert not deep.fused_supported() and deep.layers_xyz[4].in_features == 63 + 128                   # the skip of M:373
    with pytest.raises(NotImplementedError):
        full(torch.zeros(1, 90))


def test_cfgnode_roundtrip():
    import yaml
    import nerf
    src = dict(experiment=dict(id="x", train_iters=10), nerf=dict(use_viewdirs=True, train=dict(num_coarse=64, perturb=True, chunksize=2048)),
               dataset=dict(no_ndc=True, near=0.2, far=0.8), models=dict(coarse=dict(type="ConditionalBlendshapePaperNeRFModel")))
    cfg = nerf.CfgNode(src)
    assert cfg.nerf.train.num_coarse == 64 and getattr(cfg.nerf, "train").perturb is True
    assert getattr(nerf.models, cfg.models.coarse.type) is nerf.models.ConditionalBlendshapePaperNeRFModel
    assert yaml.safe_load(cfg.dump()) == src
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.dataset.near = 1.0
    c2 = cfg.clone()
    c2.defrost()
    c2.merge_from_list(["dataset.near", "0.5"])
    assert c2.dataset.near == 0.5 and cfg.dataset.near == 0.2


def test_public_api_names():
    import inspect
    import nerf
    for n in ["CfgNode", "get_embedding_function", "get_ray_bundle", "img2mse", "mse2psnr", "meshgrid_xy", "models",
              "run_one_iter_of_nerf", "load_flame_data", "load_llff_data", "dump_rays", "GaussianSmoothing",
              "positional_encoding", "sample_pdf_2", "volume_render_radiance_field", "cumprod_exclusive", "get_minibatches",
              "predict_and_render_radiance"]:
        assert hasattr(nerf, n), n
    sig = inspect.signature(nerf.run_one_iter_of_nerf)
    assert list(sig.parameters) == ["height", "width", "focal_length", "model_coarse", "model_fine", "ray_origins",
                                    "ray_directions", "options", "mode", "encode_position_fn", "encode_direction_fn",
                                    "expressions", "background_prior", "latent_code", "ray_directions_ablation"]
    assert list(inspect.signature(nerf.get_ray_bundle).parameters) == ["height", "width", "intrinsics", "tform_cam2world", "center"]
    a, b = nerf.meshgrid_xy(torch.arange(3), torch.arange(2))
    assert a.shape == (2, 3) and int(a[1, 2]) == 2 and int(b[1, 2]) == 1
    assert abs(nerf.mse2psnr(0.01) - 20.0) < 1e-9


def _flame_dataset(tmp):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synthetic_dataset as MS
    return MS.write(str(tmp), size=32, n_train=6, n_val=4, n_test=5, seed=3)


FLAME_MODES = (("full", dict()), ("half", dict(half_res=True)), ("skip2", dict(testskip=2)), ("test", dict(test=True, half_res=True)))


def _check_flame(out, get):
    imgs, poses, render_poses, hwf, i_split, expr, frontal, bboxs = out
    assert frontal is None
    assert imgs.dtype == torch.float32 and np.array_equal(imgs.numpy(), get("imgs"))
    assert poses.dtype == torch.float32 and np.array_equal(poses.numpy(), get("poses"))
    assert render_poses.numpy().dtype == get("render_poses").dtype and np.array_equal(render_poses.numpy(), get("render_poses"))
    assert [int(hwf[0]), int(hwf[1])] == get("hw").tolist()
    assert np.array_equal(np.asarray(hwf[2], dtype=np.float64), get("intrinsics"))
    assert expr.dtype == torch.float32 and np.array_equal(expr.numpy(), get("expr"))
    assert bboxs.dtype == torch.int32 and np.array_equal(bboxs.numpy(), get("bboxs"))
    for k, ix in enumerate(i_split):
        assert np.array_equal(np.asarray(ix), get(f"split{k}"))


def test_load_flame_data_against_reference_fixture(tmp_path):
    """f4: nerf.load_flame_data on the synthetic on-disk dataset equals, array for array and dtype for dtype, what the
    reference's load_flame_data (LF:40-211) returned for the same files (tests/golden/load_flame.npz; full / half_res /
    testskip / test-only), including the half_res images (cv2 INTER_AREA order, see nerf/load_flame.py:_area_resize)."""
    import nerf
    base = _flame_dataset(tmp_path)
    g = np.load(os.path.join(ROOT, "tests", "golden", "load_flame.npz"))
    for tag, kw in FLAME_MODES:
        _check_flame(nerf.load_flame_data(base, **kw), lambda k: g[f"{tag}_{k}"])


def test_load_flame_data_against_live_reference(tmp_path):
    from oracle import ref_import as RI
    if not RI.reference_available():
        pytest.skip("/root/reference only exists in the build container")
    import nerf
    base = _flame_dataset(tmp_path)
    with RI.flame_loader_io() as lf:
        for tag, kw in FLAME_MODES:
            want = lf.load_flame_data(base, **kw)
            blob = {"imgs": want[0].numpy(), "poses": want[1].numpy(), "render_poses": want[2].numpy(), "hw": np.array(want[3][:2]),
                    "intrinsics": np.asarray(want[3][2], dtype=np.float64), "expr": want[5].numpy(), "bboxs": want[7].numpy()}
            blob.update({f"split{k}": np.asarray(ix) for k, ix in enumerate(want[4])})
            _check_flame(nerf.load_flame_data(base, **kw), lambda k: blob[k])


def test_entry_points_reject_bad_arguments_before_touching_the_device(hip_lib):
    """Argument validation of the C ABI runs on the host: NULL pointers / impossible sizes must come back as NF_EINVAL (-22),
    empty work as 0, without a device call (this suite runs without a GPU)."""
    import nerf._hip as H
    lib = H.lib()
    EINVAL = -22
    assert lib.nf_ray_bundle(0, 8, 1.0, 1.0, 0.5, 0.5, None, 4, None, None, None) == EINVAL
    assert lib.nf_ray_batch(8, 8, 1.0, 1.0, 4.0, 4.0, None, 4, None, 0, 0, None, 0, None, None, None, None, None, None, None) == 0      # n = 0
    assert lib.nf_ray_batch(8, 8, 1.0, 1.0, 4.0, 4.0, None, 4, None, 0, 5, None, 0, None, None, None, None, None, None, None) == EINVAL
    assert lib.nf_sample_coarse_ex(0, 64, 0.2, 0.8, None, None, 1, None, None) == 0                                                 # no rays
    assert lib.nf_sample_coarse_ex(4, 0, 0.2, 0.8, None, None, 0, None, None) == EINVAL
    assert lib.nf_weighted_choice(None, None, 100, 0, None, None, 0, None) == 0                                                    # n = 0
    assert lib.nf_weighted_choice(None, None, 100, 10, None, None, 0, None) == EINVAL
    assert lib.nf_weighted_choice_workspace_bytes() >= 4 * (8 + 4096)
    assert lib.nf_paper_bwd_workspace_floats(2048 * 128) > 2176 * 2048 * 128
    assert lib.nf_tiny_bwd_workspace_floats(4096 * 32) > 0
    assert lib.nf_flex_mlp_bwd(4, None, None, None, 4, 4, None, 0, None, None) == EINVAL
    one = ctypes.c_void_p(16)                                                             # any non-NULL pointer: the size check comes before a launch
    assert lib.nf_flex_mlp_fwd_train(4, one, one, one, one, 1, 1 << 20, 4, one, one, None) == EINVAL      # 2^22 points: past the 32-bit section offsets
    assert lib.nf_tiny_mlp_fwd_train(one, one, one, one, 1, 1 << 20, 4, one, one, None) == EINVAL


def test_launcher_host_helpers():
    """Pure host-side pieces of the launchers: the importance map of TR:230-239 (p = 0.9 inside the bounding box, normalised) and
    the jet colour map of the error image (matplotlib's piecewise-linear 'jet' at its anchor points)."""
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    from launch import eval_sharded, train_sharded
    # the reference looks draw k (probability probs[k], probs = the row-major (H, W) map) up in coords = meshgrid_xy(arange(H),
    # arange(W)).reshape(-1, 2), i.e. pixel (k % H, k // H): the map acts transposed.  The launcher's weights are indexed by the
    # row-major pixel index row * W + col, so weight[row * W + col] must equal probs[k] of the k with coords[k] == (row, col)
    import nerf
    for H_, W_, box in ((8, 8, [2, 6, 1, 5]), (6, 10, [1, 4, 2, 9])):
        w = train_sharded.importance_maps(np.array([box]), H_, W_)[0]
        probs = np.full((H_, W_), 1 - 0.9)
        probs[box[0]:box[1], box[2]:box[3]] = 0.9
        probs = (probs / probs.sum()).reshape(-1)
        coords = torch.stack(nerf.meshgrid_xy(torch.arange(H_), torch.arange(W_)), dim=-1).reshape((-1, 2)).numpy()   # TR:302-306
        assert abs(w.sum() - 1.0) < 1e-12 and w.shape == (H_ * W_,)
        for k in range(H_ * W_):
            row, col = int(coords[k, 0]), int(coords[k, 1])
            assert w[row * W_ + col] == probs[k], (k, row, col)
    m = train_sharded.importance_maps(np.array([[2, 6, 1, 5]]), 8, 8)[0].reshape(8, 8)
    assert np.allclose(m[1:5, 2:6] / m[0, 0], 9.0) and np.count_nonzero(m == m[0, 0]) == 64 - 16     # bbox rows <-> columns
    x = torch.tensor([[0.0, 0.125, 0.375], [0.5, 0.625, 1.0]])
    got = eval_sharded.jet_u8(x).tolist()
    want = [[[0, 0, 127], [0, 0, 255], [0, 255, 255]], [[127, 255, 127], [255, 255, 0], [127, 0, 0]]]
    assert got == want, got


def test_hot_kernels_keep_their_register_and_instruction_budget(tmp_path):
    """CPU (hipcc cross-compiles gfx950): the hand-scheduled exact-f32 kernels must keep the shape the measurements were taken on --
    no spilled registers, no scratch, the 128 KiB of wave-private LDS slabs, the weight stream of the layer-streamed kernels as buffer
    loads (no 64-bit vector addresses, no flat loads), and the weight-gradient kernel's DMA in buffer form.  A compiler or flag change
    that breaks one of these costs 5-10 % of the headline without failing any numerical test."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    import build
    src_dir = os.path.join(ROOT, "4d-facial-avatars_amd", "csrc")

    def kernels(src):
        out = str(tmp_path / (src + ".s"))
        subprocess.run([hipcc, *build.FLAGS, "-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", out, os.path.join(src_dir, src)],
                       check=True, capture_output=True)
        txt = open(out).read()
        found = {}
        for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?"
                             r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
            agpr, lds, name, scratch, vgpr, spill = m.groups()
            body = re.search(r"^%s:[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(name), txt, re.S | re.M).group(1)
            found[name] = dict(lds=int(lds), scratch=int(scratch), vgpr=int(vgpr), spill=int(spill), body=body)
        return found

    def one(ks, name, tmpl):
        """the kernel whose mangled name contains `name` and the template argument `tmpl` (no assumption about the mangling's digits)"""
        hit = [v for k, v in ks.items() if re.search(r"\d+%s%s" % (re.escape(name), re.escape(tmpl)), k)]
        assert len(hit) == 1, (name, tmpl, sorted(ks))
        return hit[0]

    ks = kernels("nf_mlp.hip")
    for prefix in ("k_paper_mlp_fwd", "k_paper_mlp_fwd_save"):
        k = one(ks, prefix, "ILi2E")
        assert k["spill"] == 0 and k["scratch"] == 0 and k["vgpr"] <= 512, (prefix, k["spill"], k["scratch"], k["vgpr"])
        assert k["lds"] == 131072                                        # four wave-private 32 KiB slabs: one workgroup per CU
        b = k["body"]
        n_mfma = len(re.findall(r"v_mfma_f32_16x16x4_f32", b))            # 999,936 MFMA FLOPs per point = 31248 MFMAs per 32-point wave tile, of
        assert n_mfma == 5000, n_mfma                                    # which the K loops are rolled: exactly 5000 static instructions with this image's compiler -- a dropped or duplicated K chunk shows here
        assert "flat_load" not in b and "s_barrier" not in b
        assert len(re.findall(r"buffer_load_dwordx4", b)) >= 600          # the weight and bias stream
        assert len(re.findall(r"global_load_dwordx4", b)) == 0            # (the round-2 form: a 64-bit vector address per fragment)
    ks = kernels("nf_mlp_bwd.hip")
    k = one(ks, "k_paper_mlp_bwd_chain_masks", "ILi2E")
    assert k["spill"] == 0 and k["scratch"] == 0 and k["lds"] == 131072
    assert len(re.findall(r"buffer_load_dwordx4", k["body"])) >= 200
    k = one(ks, "k_dw_gemm_lds", "ILi0E")
    assert k["spill"] == 0 and k["scratch"] == 0 and k["vgpr"] <= 256     # eight waves = two per SIMD
    assert len(re.findall(r"buffer_load_dwordx4 .* lds", k["body"])) >= 24 and "global_load_lds" not in k["body"]


def test_branch_free_sincos_on_the_host():
    """csrc/nf_sincos.h (the in-kernel positional encoding's sin / cos pair) uses IEEE operations only -- explicit fused multiply-adds,
    round-to-nearest-even, integer bit operations -- so the device computes the bits the host computes: compile the SAME header with gcc
    and sweep it against double precision.  Gate: 1.2e-7 up to |x| = 2^20 (an encoding argument 2^9 * coordinate), 2e-7 up to 2^23,
    4e-6 up to 2^27; SURVEY 8(d)(i) asks 2e-6 of the encoding."""
    import subprocess, tempfile
    src = r'''
#include "nf_sincos.h"
#include <stdio.h>
int main(void) {
    for (int e = -24; e <= 27; ++e) {
        double ws = 0, wc = 0;
        for (int i = 0; i < 400000; ++i) {
            float x = ldexpf(1.0f + (float)i / 400000.0f, e) * ((i & 1) ? -1.f : 1.f), s, c;
            nf_sincos(x, &s, &c);
            double es = fabs((double)s - sin((double)x)), ec = fabs((double)c - cos((double)x));
            if (es > ws) ws = es;
            if (ec > wc) wc = ec;
        }
        printf("%d %.4e %.4e\n", e, ws, wc);
    }
    float s, c; nf_sincos(0.0f, &s, &c); printf("zero %g %g\n", s, c);
    return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "t.c"), "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "4d-facial-avatars_amd", "csrc"), "-o", exe, os.path.join(td, "t.c"), "-lm"],
                       check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    rows = [l.split() for l in out if l and not l.startswith("zero")]
    assert len(rows) == 52
    for e, ws, wc in rows:
        gate = 1.2e-7 if int(e) < 20 else (2e-7 if int(e) < 23 else 4e-6)
        assert float(ws) <= gate and float(wc) <= gate, (e, ws, wc)
    assert [l for l in out if l.startswith("zero")] == ["zero 0 1"]


def test_no_kernel_of_the_library_spills(hip_lib):
    """Every kernel of libnerface_hip.so, as the compiler reports it when the library is built (-Rpass-analysis=kernel-resource-usage,
    kept per translation unit in lib/obj/<unit>.usage.txt): no spilled vector OR scalar registers, no scratch memory, at most 512 VGPRs + AGPRs."""
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    import build
    obj = os.path.join(build.OUT_DIR, "obj")
    units = [s.replace(".hip", ".usage.txt") for s in build.SOURCES]
    if not all(os.path.exists(os.path.join(obj, u)) for u in units):
        build.build(force=True, verbose=False)               # objects cached from before the remarks were kept
    n = 0
    for u in units:
        name = None
        for ln in open(os.path.join(obj, u)):
            m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill):\s+(\S+)", ln)
            if not m:
                continue
            key, val = m.groups()
            if key == "Function Name":
                name, n = val, n + 1
            elif key in ("ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill"):
                # round 5: scalar spills too (round 4 had 38 in the headline kernel -- 32 hoisted lane masks of the per-slot coordinate
                # selection of the positional encoding, not, as DESIGN r04 said, sincosf's branches -- and 29 in k_grad_reduce)
                assert int(val) == 0, (u, name, key, val)
            elif key in ("VGPRs", "AGPRs"):
                assert int(val) <= 512, (u, name, key, val)
    assert n >= 70, n                                           # 79 kernels in round 3
