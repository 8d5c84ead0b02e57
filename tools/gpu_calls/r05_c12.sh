# round 5, calls 12+: nothing but the driver's exact command on one more fresh box (the slow-store condition of the driver's boxes showed on
# 2 of 2 of ITS boxes and 1 of ~45 of the builder's: every further box is a sample; the line now carries what would explain one)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c12_$1; mkdir -p $O
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc $?"
tail -n 1 $O/bench.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['summary']; print({k: s.get(k) for k in ('value_rays_s','roofline_frac','split_f16x2_rays_s','train_ms_per_iter_f32','train_ms_per_iter_f16x3','train_ms_per_iter_bf16x3','train_bf16x3_fwd_save_ms','train_bf16x3_fwd_save_ms_at_2400mhz','pattern_store_gbs','pattern_store_default_policy_gbs','power_w_train_fwd_bf16x3','sclk_mhz_train_fwd_bf16x3','launch_ms_train_fwd_bf16x3','power_w_f16x3','sclk_mhz_f16x3')})"
