#!/bin/bash
# usage: r5_ab_env_cfg.sh CONFIG VAR A B [pairs] [steps]  - interleaved bench pairs with VAR=A / VAR=B on a named config
C=$1; V=$2; A=$3; B=$4; N=${5:-3}; S=${6:-100}
run() { env $V=$1 python3 bench.py --config $C --steps $S --warmup 15 --no-cpu-baseline --no-secondary --no-dp1 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq $N); do echo "$C $V=$A $(run $A)   $V=$B $(run $B)"; done
