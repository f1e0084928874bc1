for cfg in "1 4" "1 8" "2 8"; do
  set -- $cfg
  DOSX_WGRAD_NT=$1 DOSX_WGRAD_MAXSPLIT=$2 python tools/wgrad_saturated.py 2>&1 | grep "^NT"
done
python tools/bench_wgroup.py 2>&1 | tail -6
python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['roofline']['avg_us'], r['roofline']['frac'], r['secondary'])"
