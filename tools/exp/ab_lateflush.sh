for v in 1 0 1 0 1 0 1 0; do
  export DOSX_SPLIT_LATE_FLUSH=$v
  python bench.py --no-secondary --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('cfg2 late flush', os.environ['DOSX_SPLIT_LATE_FLUSH'], r['ms_per_step'])"
done
for v in 1 0 1 0; do
  export DOSX_SPLIT_LATE_FLUSH=$v
  python bench.py --config edos_h256_b64 --no-secondary --no-cpu-baseline --steps 60 --warmup 16 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('edos late flush', os.environ['DOSX_SPLIT_LATE_FLUSH'], r['ms_per_step'])"
done
