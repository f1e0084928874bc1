# re-tune of the grouped weight gradients' split cap and the late-flush policy after the tail splits went their own way
for v in 8 6 12 8 6 12; do
  export DOSX_WGRAD_MAXSPLIT=$v
  python bench.py --no-secondary --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('cfg2 maxsplit', os.environ['DOSX_WGRAD_MAXSPLIT'], r['ms_per_step'])"
done
unset DOSX_WGRAD_MAXSPLIT
for v in 1 0 2 1 0 2; do
  export DOSX_SPLIT_LATE_FLUSH=$v
  python bench.py --no-secondary --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('cfg2 late flush', os.environ['DOSX_SPLIT_LATE_FLUSH'], r['ms_per_step'])"
done
unset DOSX_SPLIT_LATE_FLUSH
for v in 8 6 12 8 6 12; do
  export DOSX_WGRAD_MAXSPLIT=$v
  python bench.py --config edos_h256_b64 --no-secondary --no-cpu-baseline --steps 60 --warmup 16 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('edos maxsplit', os.environ['DOSX_WGRAD_MAXSPLIT'], r['ms_per_step'])"
done
