ms() { python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['ms_per_step'], r.get('host_ms_per_step'))"; }
for rep in 1 2 3; do
  echo -n "HEAD dp1: "; (cd $GRAFT_REPO_ROOT && python3 bench.py --force-dist --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms)
  echo -n "old  dp1: "; (cd $GRAFT_REPO_ROOT/_old && python3 bench.py --force-dist --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms)
  echo -n "HEAD plain: "; (cd $GRAFT_REPO_ROOT && python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms)
  echo -n "old  plain: "; (cd $GRAFT_REPO_ROOT/_old && python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | ms)
done
