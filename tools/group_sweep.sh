for g in ${GROUPS_TO_TRY:-2 4 8 16}; do
  echo "group=$g"; timeout -k 10 120 python bench.py --cpu-pairs 0 --no-configs --group $g --steps 5 --warmup 2 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])" || exit 1
done
