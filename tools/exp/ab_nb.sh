for nb in 3 2 3 2 3 2; do
  export DOSX_WGRAD_NB=$nb
  python bench.py --no-secondary --no-cpu-baseline --steps 200 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('nb', os.environ['DOSX_WGRAD_NB'], r['ms_per_step'], r['roofline']['avg_us'])"
done
