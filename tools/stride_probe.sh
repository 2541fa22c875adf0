# does the row stride matter at 4K? same height, widths around 3840 (run on the GPU box)
for w in ${WIDTHS:-3840 3904 3968 4096}; do
  timeout -k 10 200 python bench.py --cpu-pairs 0 --no-configs --width $w --height 2160 --levels 5 --batch 16 --steps 3 --warmup 1 > /tmp/sp.json || exit 1
  W=$w python - <<'PY'
import json, os
w = int(os.environ["W"]); d = json.load(open("/tmp/sp.json")); k = d["roofline"]["all_kernels_ms"]
print(w, d["value"], "L0 sweeps ms per Mpx: %.4f" % (k["blur_iter"] / (16 * w * 2160 / 1e6)), k)
PY
done
