#!/bin/bash
# A/B of the ticket's memory order (common.h DOSX_TICKET_ORDER): relaxed + per-access sc1 (shipped) vs __ATOMIC_ACQ_REL.
# Interleaved bench runs + the weight-gradient group microbenchmark + the stress tests on the ACQ_REL build.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
make -C dostransformer_amd/csrc acqrel > /dev/null 2>&1 || { echo "acqrel build failed"; exit 1; }
A=$(pwd)/dostransformer_amd/csrc/build/libdosx_acqrel.so
for i in 1 2 3; do
  echo -n "relaxed cfg2: "; python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  echo -n "acq_rel cfg2: "; DOSX_LIB=$A python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
for i in 1 2; do
  echo -n "relaxed edos: "; python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
  echo -n "acq_rel edos: "; DOSX_LIB=$A python3 bench.py --config edos_h256_b64 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
echo "--- groups alone, relaxed"; python3 tools/bench_wgroup.py 2>/dev/null | tail -6
echo "--- groups alone, acq_rel"; DOSX_LIB=$A python3 tools/bench_wgroup.py 2>/dev/null | tail -6
echo "--- stress tests on the acq_rel build"
DOSX_LIB=$A python3 -m pytest tests/test_gpu_round4.py -q -k stress 2>&1 | tail -3
