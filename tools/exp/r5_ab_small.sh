#!/bin/bash
# small shapes: factored one-launch message-passing layer on / off (phonon H64 B8 = configs[0]; eDOS H256 t4 b32)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
for i in 1 2 3; do
for v in 100 0; do
DOSX_FACTOR_MIN_GF=$v timeout 300 python3 bench.py --config phonon_h64_b8 --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('h64_b8 min_gf=$v ', d['ms_per_step'])"
done
done
for i in 1 2; do
for v in 4 0.5; do
DOSX_FACTOR_MIN_GF=$v timeout 300 python3 bench.py --config edos_h256_t4_b32 --steps 30 --warmup 8 --no-cpu-baseline --no-secondary 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('edos_t4_b32 min_gf=$v ', d['ms_per_step'])"
done
done
