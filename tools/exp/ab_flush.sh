for v in 0 1 0 1 0 1; do
  export DOSX_LATE_SELF_FLUSH=$v
  python bench.py --no-secondary --no-cpu-baseline --steps 200 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('phonon late_self_flush', os.environ['DOSX_LATE_SELF_FLUSH'], r['ms_per_step'])"
done
for v in 0 1 0 1; do
  export DOSX_LATE_SELF_FLUSH=$v
  python bench.py --config edos_h256_b64 --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('edos late_self_flush', os.environ['DOSX_LATE_SELF_FLUSH'], r['ms_per_step'])"
done
