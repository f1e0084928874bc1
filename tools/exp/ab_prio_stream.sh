# (needs the DOSX_BENCH_MAIN_PRIO knob of the experiment in bench.py: run_workload inside torch.cuda.stream(Stream(priority=-1)))
# A/B: the step's main chain on a high-priority HIP stream (weight gradients stay at normal priority), with and without
# shorter-lived weight-gradient workgroups (DOSX_WGRAD_MAXCHUNKS)
for cfg in "0 0" "1 0" "1 50" "1 25" "0 0" "1 0" "1 50" "1 25"; do
  set -- $cfg
  if [ "$1" = "1" ]; then export DOSX_BENCH_MAIN_PRIO=1; else unset DOSX_BENCH_MAIN_PRIO; fi
  export DOSX_WGRAD_MAXCHUNKS=$2
  python bench.py --config edos_h256_b64 --no-secondary --no-cpu-baseline --steps 60 --warmup 16 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('edos prio', os.environ.get('DOSX_BENCH_MAIN_PRIO','0'), 'maxchunks', os.environ['DOSX_WGRAD_MAXCHUNKS'], r['ms_per_step'])"
done
for cfg in "0 0" "1 0" "1 16" "0 0" "1 0" "1 16"; do
  set -- $cfg
  if [ "$1" = "1" ]; then export DOSX_BENCH_MAIN_PRIO=1; else unset DOSX_BENCH_MAIN_PRIO; fi
  export DOSX_WGRAD_MAXCHUNKS=$2
  python bench.py --no-secondary --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys,os; r=json.loads(sys.stdin.read()); print('cfg2 prio', os.environ.get('DOSX_BENCH_MAIN_PRIO','0'), 'maxchunks', os.environ['DOSX_WGRAD_MAXCHUNKS'], r['ms_per_step'])"
done
