for ms in 8 4 2 8 4; do
  export DOSX_WGRAD_MAXSPLIT=$ms
  python bench.py --config edos_h256_b64 --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('edos maxsplit', os.environ['DOSX_WGRAD_MAXSPLIT'], r['ms_per_step'], r['roofline']['avg_us'])"
done
