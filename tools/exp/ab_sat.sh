for cfg in "1 4" "1 8" "1 16" "2 4" "2 8" "2 16" "2 32"; do
  set -- $cfg
  DOSX_WGRAD_NT=$1 DOSX_WGRAD_MAXSPLIT=$2 python tools/wgrad_saturated.py 2>&1 | grep "^NT"
done
