#!/bin/bash
# ONE parametrised script for the GPU-box calls of a round (replaces the 85 one-off tools/gpu_calls/c*.sh of rounds 3-5):
#     gpurun --timeout T -- 'bash tools/gpu_call.sh <tag> <recipe> [args] [-- <recipe> [args]] ...'
# Output of every recipe goes to gpurun_out/<tag>/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
# Recipes:
#   suite [pytest args]        pytest -m gpu (default: the whole suite)               -> pytest_gpu.txt
#   smoke                      __graft_entry__.smoke()                                 -> smoke.txt
#   gate                       tests/test_gpu_gate.py in measure mode, JSON dumps      -> gate.txt, gate.*.json
#   sweep [frames]             tools/frame_gate_sweep.py                               -> gate_sweep.txt / .json
#   ab <variant> [notime]      tools/split_fwd_ab.py on the default library and on lib/libnerface_hip_<variant>.so, hashes diffed -> ab_<variant>.txt
#   pmcmix <prec> [variant]    instruction-mix PMC passes (3 passes, 8 SQ counters each) over tools/pmc_one_launch.py <prec> -> pmc_<prec>[_variant].md
#   pmcset <prec> <variant|-> <counters...>  one PMC pass with the named counters over tools/pmc_one_launch.py -> pmcset_<prec>[_variant].md
#   pmctrain <prec> [variant]  the same passes over tools/pmc_train_launch.py <prec>   -> pmc_train_<prec>[_variant].md
#   bench [bench.py args]      python bench.py ...                                     -> bench_line.json (last line), bench.log
#   stats [bench.py args]      rocprofv3 --kernel-trace --stats of bench.py --no-extras --no-cpu-baseline ... -> kernel_stats.md
#   abpy <variant> <script> [args]  python <script> ... on the default library and on the variant, outputs side by side -> <script>_{default,variant}.txt
#   py <script> [args]         python <script> ... (cwd = repo root)                   -> <script basename>.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
ROOT=$PWD
TAG=$1; shift
O=$ROOT/gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp
L=$ROOT/4d-facial-avatars_amd/lib
PMC1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"
PMC2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PMC3="GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES"

pmc_passes() {   # $1 = output stem, $2 = launch script, $3 = precision, $4 = variant or ""
  local stem=$1 script=$2 prec=$3 var=$4 lib=""
  [ -n "$var" ] && lib=$L/libnerface_hip_$var.so
  local dbs=""
  for i in 1 2 3; do
    local grp; eval grp=\$PMC$i
    rm -rf /tmp/pmc_$i
    ( cd /tmp && NERFACE_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$i -o p -- python $ROOT/tools/$script $prec > $O/$stem.pass$i.log 2>&1 ); echo "pmc pass $i rc=$?"
    dbs="$dbs $(find /tmp/pmc_$i -name '*.db')"
  done
  python tools/rocpd_summary.py pmc $dbs > $O/$stem.md 2>&1
  grep -E "mlp_fwd|chain|dw_gemm" $O/$stem.md | cut -c1-170 | head -60
}

run_recipe() {
  local r=$1; shift
  case $r in
    suite) timeout 1500 python -m pytest tests -m gpu -q -x "$@" > $O/pytest_gpu_full.txt 2>&1; grep -v Warning $O/pytest_gpu_full.txt | tail -25 | tee $O/pytest_gpu.txt ;;
    smoke) timeout 300 python -c "import __graft_entry__ as G; G.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt ;;
    gate)  NERFACE_GATE_MEASURE=1 NERFACE_GATE_JSON=$O/gate timeout 900 python -m pytest tests/test_gpu_gate.py -m gpu -q -s "$@" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tee $O/gate.txt ;;
    sweep) timeout 900 python tools/frame_gate_sweep.py ${1:-8} $O/gate_sweep.json 2>&1 | tee $O/gate_sweep.txt | tail -12 ;;
    ab)    local v=$1; shift
           timeout 600 python tools/split_fwd_ab.py "$@" > $O/ab_default.txt 2>&1
           NERFACE_HIP_LIB=$L/libnerface_hip_$v.so timeout 600 python tools/split_fwd_ab.py "$@" > $O/ab_$v.txt 2>&1
           echo "== hashes that differ between default and $v:"; diff <(grep ^hash $O/ab_default.txt) <(grep ^hash $O/ab_$v.txt) | head -40
           echo "== times (default | $v):"; paste -d'|' <(grep ^time $O/ab_default.txt) <(grep ^time $O/ab_$v.txt | sed 's/^time [a-z0-9 ]*: //') ;;
    pmcmix)   pmc_passes "pmc_$1${2:+_$2}" pmc_one_launch.py "$1" "$2" ;;
    pmcset)   # pmcset <prec> <variant or -> <counter> [counter ...]: one extra PMC pass with the given counters -> pmcset_<prec>[_variant].md
              local prec=$1 var=$2; shift 2; [ "$var" = "-" ] && var=""
              local lib=""; [ -n "$var" ] && lib=$L/libnerface_hip_$var.so
              rm -rf /tmp/pmc_x
              ( cd /tmp && NERFACE_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_x -o p -- python $ROOT/tools/pmc_one_launch.py $prec > $O/pmcset_$prec${var:+_$var}.log 2>&1 ); echo "pmcset rc=$?"
              python tools/rocpd_summary.py pmc $(find /tmp/pmc_x -name '*.db') > $O/pmcset_$prec${var:+_$var}.md 2>&1
              grep -E "mlp_fwd|counter|^\|" $O/pmcset_$prec${var:+_$var}.md | cut -c1-200 | head -20; tail -3 $O/pmcset_$prec${var:+_$var}.log ;;
    pmctrain) pmc_passes "pmc_train_$1${2:+_$2}" pmc_train_launch.py "$1" "$2" ;;
    bench) timeout 1500 python bench.py "$@" > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log > $O/bench_line.json; cp gpurun_out/bench_detail.json $O/ 2>/dev/null; tail -c 2500 $O/bench_line.json ;;
    stats) rm -rf /tmp/stats; ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/stats -o s -- python $ROOT/bench.py --no-extras --no-cpu-baseline "$@" > $O/stats.log 2>&1 ); echo "stats rc=$?"
           python tools/rocpd_summary.py stats $(find /tmp/stats -name '*.db' | head -1) > $O/kernel_stats.md 2>&1; head -14 $O/kernel_stats.md | cut -c1-150 ;;
    abpy)  local v=$1 s=$2; shift 2; local b=$(basename $s .py)
           timeout 600 python "$s" "$@" > $O/${b}_default.txt 2>&1
           NERFACE_HIP_LIB=$L/libnerface_hip_$v.so timeout 600 python "$s" "$@" > $O/${b}_$v.txt 2>&1
           echo "== $s: default | $v"; paste -d'|' <(grep -v Warning $O/${b}_default.txt | cut -c1-110) <(grep -v Warning $O/${b}_$v.txt | cut -c1-110) | tail -14 ;;
    py)    local s=$1; shift; timeout 900 python "$s" "$@" 2>&1 | tee $O/$(basename $s .py).txt | tail -40 ;;
    *) echo "unknown recipe $r"; return 2 ;;
  esac
}

args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then run_recipe "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_recipe "${args[@]}"
exit 0
