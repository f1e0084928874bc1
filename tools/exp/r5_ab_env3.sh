#!/bin/bash
# usage: r5_ab_env3.sh VAR A B C [rounds]
V=$1; A=$2; B=$3; Cc=$4; N=${5:-3}
run() { env $V=$1 python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary --no-dp1 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq $N); do echo "$V=$A $(run $A)   $V=$B $(run $B)   $V=$Cc $(run $Cc)"; done
