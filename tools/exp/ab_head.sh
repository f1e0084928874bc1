# A/B: normalised head outputs from the heads' GEMM epilogues (1) vs a dosx_rownorm launch behind them (0)
for v in 1 0 1 0 1 0; do
  export DOSX_FUSED_HEAD_NORM=$v
  python bench.py --no-secondary --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('fused', os.environ['DOSX_FUSED_HEAD_NORM'], r['ms_per_step'])"
done
