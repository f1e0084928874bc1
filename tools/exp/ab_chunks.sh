for mc in 0 50 30 20 0 30; do
  export DOSX_WGRAD_MAXCHUNKS=$mc
  python bench.py --config edos_h256_b64 --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('edos maxchunks', os.environ['DOSX_WGRAD_MAXCHUNKS'], r['ms_per_step'], r['roofline']['avg_us'])"
done
for mc in 0 30 20 0 30 20; do
  export DOSX_WGRAD_MAXCHUNKS=$mc
  python bench.py --no-secondary --no-cpu-baseline --steps 200 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('phonon maxchunks', os.environ['DOSX_WGRAD_MAXCHUNKS'], r['ms_per_step'], r['roofline']['avg_us'])"
done
