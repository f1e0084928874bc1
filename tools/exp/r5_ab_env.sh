#!/bin/bash
# usage: r5_ab_env.sh VAR A B [pairs]   - interleaved bench pairs with VAR=A / VAR=B
V=$1; A=$2; B=$3; N=${4:-3}
run() { env $V=$1 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --no-dp1 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq $N); do echo "$V=$A $(run $A)   $V=$B $(run $B)"; done
