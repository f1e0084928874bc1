# A/B: 128x64 (NT=2) vs 64x64 (NT=1) weight-gradient tiles, and the split cap with the larger tile
for cfg in "2 8" "1 8" "2 16" "2 8" "1 8" "2 16" "2 12"; do
  set -- $cfg
  export DOSX_WGRAD_NT=$1 DOSX_WGRAD_MAXSPLIT=$2
  python bench.py --no-secondary --no-cpu-baseline --steps 200 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('nt', os.environ['DOSX_WGRAD_NT'], 'maxsplit', os.environ['DOSX_WGRAD_MAXSPLIT'], r['ms_per_step'], r['roofline']['site'], r['roofline']['avg_us'])"
done
