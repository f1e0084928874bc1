mkdir -p gpurun_out/exp
run() { echo "== $1 $3"; env $1 python bench.py --steps 200 --no-cpu-baseline $3 2> gpurun_out/exp/err_$2.txt | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'], r['roofline']['frac'])"; }
run "X=1" a
run "HIP_FORCE_DEV_KERNARG=1" b
run "HIP_FORCE_DEV_KERNARG=0" c
run "X=1" d "--launch graph"
run "HIP_FORCE_DEV_KERNARG=1" e "--launch graph"
run "GPU_MAX_HW_QUEUES=8" f
run "HSA_ENABLE_SDMA=0" g
