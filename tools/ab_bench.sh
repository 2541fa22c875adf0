#!/bin/bash
# A/B of tuning options on one box: every variant is one short bench run (no CPU legs, no extra configs); prints pairs/s and the per-class kernel times.
# usage: tools/ab_bench.sh "name=val name2=val" "name=other" ...      (an empty string = defaults; names are mav_set_option options)
set -u
for v in "$@"; do
  echo "== ${v:-defaults}"
  opts=""; for kv in $v; do opts="$opts --opt $kv"; done
  timeout -k 10 200 python bench.py --steps ${STEPS:-30} --cpu-pairs 0 --no-verify --no-configs $opts ${BENCH_ARGS:-} | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])" || exit 1
done
