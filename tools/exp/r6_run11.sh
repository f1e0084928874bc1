#!/bin/bash
# Electron-DOS step with the plain feed-forward GEMMs on the split-bf16 kernel (opt-in DOSX_FFN_BF16X3): interleaved A/B
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O="$R/gpurun_out/r6_run11"
mkdir -p "$O"
cd "$R"
python -m pytest tests/test_gpu_gemm.py -q -m gpu -k bf16x3 2>&1 | tail -2
run() {
  name=$1; cfg=$2; steps=$3; shift 3
  env "$@" timeout 300 python bench.py --allow-env --no-secondary --no-cpu-baseline --config $cfg --steps $steps 2> /dev/null | \
    python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$name', '$cfg', r['ms_per_step'], r['check']['loss_first'], r['check']['loss_last'], r['check']['replay_eq_eager'])" | tee -a "$O/ab.log"
}
for i in 1 2 3; do
  run "exact fp32            " edos_h256_b64 40 DOSX_X=0
  run "split-bf16 ffn        " edos_h256_b64 40 DOSX_FFN_BF16X3=1
  run "split-bf16 ffn, notail" edos_h256_b64 40 DOSX_FFN_BF16X3=1 DOSX_FFN_TAIL=0
  run "exact fp32            " edos_h256_t4_b32 40 DOSX_X=0
  run "split-bf16 ffn        " edos_h256_t4_b32 40 DOSX_FFN_BF16X3=1
done
