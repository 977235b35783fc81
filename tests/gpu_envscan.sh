#!/bin/bash
# development aid: the overlapped step at c2 for values of one environment tunable
# usage: gpu_envscan.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null |
    python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$VAR=$v', round(d['ms_per_step'],3), {k: round(x, 2) for k, x in d['phases_ms_rank0'].items()})
" || exit 1
done
