mkdir -p gpurun_out/exp
run() { echo "== $1"; env $1 python bench.py --steps 300 --no-cpu-baseline 2> gpurun_out/exp/err_$2.txt | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"; }
run "X=1" a
run "DOSX_FFN_HALF_MAX=256 DOSX_FFN_KB=32" b
run "DOSX_FFN_HALF_MAX=256" c
run "DOSX_FFN_KB=32" d
run "X=1" e
