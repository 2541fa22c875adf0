#!/bin/bash
# Everything profiles/<round>/ holds, measured on the box this runs on:   tools/profile_round.sh <outdir>
#   kernel-trace stats of the default bench (1080p, batch 64), of the 4K / 5-layer shape and of BASELINE config 2 (1280x720, one pair) --
#   each from the HEADLINE LOOP ONLY (--no-profile --no-configs --no-api-loop --no-verify: no video leg, no calibration probe, no second
#   schedule), so that launches per step and per-class times can be read off the table directly --, PMC passes of all three, traffic
#   records, the sweeps' busy-time unions, the step anatomy, and one config-2 call launch by launch.
set -u
export TMPDIR=/tmp
out=$1; mkdir -p "$out"
B4K="--width 3840 --height 2160 --levels 5 --batch 16"
CLEAN="--cpu-pairs 0 --no-configs --no-verify --no-profile --no-api-loop"
# untraced records first: the algorithmic bytes per sweep launch (roofline block) the unions below are quoted on
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-configs --no-api-loop > "$out/untraced1080.json" 2> "$out/untraced1080.err" || exit 1
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --cpu-pairs 0 --no-configs --no-api-loop $B4K > "$out/untraced4k.json" 2> "$out/untraced4k.err" || exit 1
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --cpu-pairs 0 --no-configs --no-api-loop --width 1280 --height 720 --batch 1 > "$out/untraced720.json" 2> "$out/untraced720.err" || exit 1
# 5 + 2 steps each: the stats table then holds 7 identical steps and nothing else
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt1080" -- python3 bench.py --steps 5 --warmup 2 $CLEAN > "$out/kt1080.json" 2> "$out/kt1080.err" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt4k" -- python3 bench.py --steps 5 --warmup 2 $CLEAN $B4K > "$out/kt4k.json" 2> "$out/kt4k.err" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt720" -- python3 bench.py --steps 50 --warmup 10 $CLEAN --width 1280 --height 720 --batch 1 > "$out/kt720.json" 2> "$out/kt720.err" || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d "$out/ktc2" -- python3 tools/c2_probe.py 1280 720 1 50 > "$out/ktc2.log" 2>&1 || exit 1
tools/pmc_passes.sh "$out/pmc1080" --batch 64 > "$out/pmc1080.log" 2>&1 || exit 1
tools/pmc_passes.sh "$out/pmc4k" $B4K > "$out/pmc4k.log" 2>&1 || exit 1
tools/pmc_passes.sh "$out/pmc720" --width 1280 --height 720 --batch 1 > "$out/pmc720.log" 2>&1 || exit 1
python3 tools/make_traffic.py "$out/pmc1080" "$out/traffic.json" --batch 64 > /dev/null || exit 1
python3 tools/make_traffic.py "$out/pmc4k" "$out/traffic_4k.json" --width 3840 --height 2160 --levels 5 --batch 16 > /dev/null || exit 1
python3 tools/make_traffic.py "$out/pmc720" "$out/traffic_720p.json" --width 1280 --height 720 --levels 1 --batch 1 > /dev/null || exit 1
# sweep launches overlap (two pairs in flight): busy time = union of the dispatch intervals of the kernel trace
for t in kt1080 kt4k kt720; do
  u=${t/kt/untraced}
  per=$(python3 -c "import json,sys; print(json.loads([l for l in open('$out/$u.json') if l.startswith('{')][-1])['roofline']['alg_bytes_per_launch_avg'])")
  python3 tools/trace_union.py "$out"/$t/runc/*_kernel_trace.csv --bytes-per-dispatch "$per" > "$out/sweep_busy_$t.txt" || exit 1
  # the step time of the TRACED run itself (its own bench line) and of the untraced one: bench.py quotes both next to the two rates
  python3 -c "import json; t=json.loads([l for l in open('$out/$t.json') if l.startswith('{')][-1]); u=json.loads([l for l in open('$out/$u.json') if l.startswith('{')][-1]); print(f\"  traced step  {t['ms_per_step']:.3f} ms per step ({t['steps']} timed steps under rocprofv3 --kernel-trace); untraced {u['ms_per_step']:.3f} ms\")" >> "$out/sweep_busy_$t.txt" || exit 1
  [ "$t" = kt720 ] || python3 tools/step_anatomy.py "$out"/$t/runc/*_kernel_trace.csv 3 > "$out/step_anatomy_$t.txt" || exit 1
done
python3 tools/trace_timeline.py "$out"/ktc2/runc/*_kernel_trace.csv --last 27 > "$out/c2_timeline.txt" || exit 1
# the same anatomy WITHOUT a tracer (HIP events around runs of launches / around every launch) and the layer-image kernels per launch
python3 tools/untraced_anatomy.py 1920 1080 64 1 > "$out/untraced_anatomy_1080p_b64.txt" 2>&1 || exit 1
python3 tools/untraced_anatomy.py 3840 2160 16 5 > "$out/untraced_anatomy_4k_l5_b16.txt" 2>&1 || exit 1
python3 tools/blur_launches.py "$out"/kt1080/runc/*_kernel_trace.csv > "$out/layer_image_launches_1080p_b64.txt" || exit 1
python3 tools/blur_launches.py "$out"/kt4k/runc/*_kernel_trace.csv > "$out/layer_image_launches_4k_l5_b16.txt" || exit 1
# what travels back: the summaries.  The raw dispatch traces and counter CSVs (> 64 MB together) stay on the box.
for t in kt1080 kt4k kt720; do cp "$out"/$t/runc/*_kernel_stats.csv "$out/kernel_stats_$t.csv"; done
for t in pmc1080 pmc4k pmc720; do cp "$out/$t/summary.txt" "$out/pmc_summary_$t.txt"; done
rm -rf "$out"/kt1080 "$out"/kt4k "$out"/kt720 "$out"/ktc2 "$out"/pmc1080 "$out"/pmc4k "$out"/pmc720
du -sh "$out"; ls "$out"
