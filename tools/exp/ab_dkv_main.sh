#!/bin/bash
# A/B: the streamed dK/dV kernels (more than 64 keys) on the side stream (default) or on the main stream
# (knob removed after the measurement - profiles/r04_ab_dkv_main.log: no difference; in functional.encoder_bwd take the `sink.join(); ops.attention_bwd(a2)` branch unconditionally to repeat it)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'])"; }
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "edos dkv_main=$v: "; DOSX_DKV_MAIN=$v python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | ms
    echo -n "edos_t4_b32 dkv_main=$v: "; DOSX_DKV_MAIN=$v python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | ms
  done
done
