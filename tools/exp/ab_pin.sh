for v in 0 64 0 64; do
  export DOSX_PIN_RING=$v
  python bench.py --shuffle --no-secondary --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('shuffle pin_ring', os.environ['DOSX_PIN_RING'], r['ms_per_step'], r['host_ms_per_step'], r['host_loop_ms_per_step'])"
done
