for v in 0 1 0 1 0 1; do
  if [ $v = 1 ]; then export DOSX_DEBUG_SKIP_COMBINE=1; else unset DOSX_DEBUG_SKIP_COMBINE; fi
  python bench.py --no-secondary --no-cpu-baseline --steps 200 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('skip_combine', os.environ.get('DOSX_DEBUG_SKIP_COMBINE'), r['ms_per_step'])"
done
