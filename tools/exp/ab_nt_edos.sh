# A/B on the eDOS configuration: 128x64 (NT=2) vs 64x64 (NT=1) weight-gradient tiles
for cfg in "1 8" "2 8" "2 16" "1 8" "2 8" "2 16"; do
  set -- $cfg
  export DOSX_WGRAD_NT=$1 DOSX_WGRAD_MAXSPLIT=$2
  python bench.py --config edos_h256_b64 --no-secondary --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('nt', os.environ['DOSX_WGRAD_NT'], 'maxsplit', os.environ['DOSX_WGRAD_MAXSPLIT'], r['ms_per_step'], r['roofline']['site'], r['roofline']['avg_us'], r['roofline']['frac'])"
done
