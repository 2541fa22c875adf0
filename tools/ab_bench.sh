#!/bin/bash
# A/B of tuning switches on one box: every variant is one short bench run (no CPU legs); prints pairs/s and the per-class kernel times.
# usage: tools/ab_bench.sh "VAR=val VAR2=val" "VAR=other" ...      (an empty string = defaults)
set -u
for v in "$@"; do
  echo "== ${v:-defaults}"
  env $v timeout -k 10 200 python bench.py --steps 30 --cpu-pairs 0 --no-verify ${BENCH_ARGS:-} | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])" || exit 1
done
