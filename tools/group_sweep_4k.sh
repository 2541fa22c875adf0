# group sweep on the 4K / 5-layer preset (run on the GPU box)
for g in ${GROUPS_TO_TRY:-1 2 4 8 16}; do
  echo "group=$g"; timeout -k 10 200 python bench.py --cpu-pairs 0 --no-configs --width 3840 --height 2160 --levels 5 --batch 16 --steps 3 --warmup 1 --group $g | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])" || exit 1
done
