#!/bin/bash
# same-box A/B of two builds of the library: build/libdosx_prev.so (git archive of the previous commit) against the in-tree one
# usage: r5_ab_lib.sh [config] [pairs] [steps]
C=${1:-phonon_h128_b64}; N=${2:-3}; S=${3:-300}
run() { env $1 python3 bench.py --config $C --steps $S --warmup 30 --no-cpu-baseline --no-secondary --no-dp1 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq $N); do echo "$C prev $(run DOSX_LIB=dostransformer_amd/csrc/build/libdosx_prev.so)   new $(run DOSX_NOP=1)"; done
