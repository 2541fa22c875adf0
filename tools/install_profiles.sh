#!/bin/bash
# Copy the summaries tools/profile_round.sh left under <src> into profiles/<round>/ under their committed names.
#   tools/install_profiles.sh gpurun_out/r5prof3 profiles/r05
set -eu
R=$1; P=$2; mkdir -p "$P"
cp "$R/kernel_stats_kt1080.csv" "$P/kernel_stats_1080p_b64.csv"
cp "$R/kernel_stats_kt4k.csv" "$P/kernel_stats_4k_l5_b16.csv"
cp "$R/kernel_stats_kt720.csv" "$P/kernel_stats_720p_b1.csv"
cp "$R/pmc_summary_pmc1080.txt" "$P/pmc_summary_1080p_b64.txt"
cp "$R/pmc_summary_pmc4k.txt" "$P/pmc_summary_4k_l5_b16.txt"
cp "$R/pmc_summary_pmc720.txt" "$P/pmc_summary_720p_b1.txt"
cp "$R/traffic.json" "$R/traffic_4k.json" "$R/traffic_720p.json" "$P/"
cp "$R/sweep_busy_kt1080.txt" "$P/sweep_busy_1080p_b64.txt"
cp "$R/sweep_busy_kt4k.txt" "$P/sweep_busy_4k_l5_b16.txt"
cp "$R/sweep_busy_kt720.txt" "$P/sweep_busy_720p_b1.txt"
cp "$R/step_anatomy_kt1080.txt" "$P/step_anatomy_1080p_b64.txt"
cp "$R/step_anatomy_kt4k.txt" "$P/step_anatomy_4k_l5_b16.txt"
cp "$R"/untraced_anatomy_*.txt "$R"/layer_image_launches_*.txt "$P/"
cp "$R/c2_timeline.txt" "$P/c2_timeline_720p_b1.txt"
cp "$R/kt1080.json" "$P/bench_under_kernel_trace_1080p.json"
cp "$R/kt4k.json" "$P/bench_under_kernel_trace_4k.json"
cp "$R/kt720.json" "$P/bench_under_kernel_trace_720p.json"
cp "$R/untraced4k.json" "$P/bench_4k_l5_b16.json"
round=$(basename "$P")
sed -i "s#\"source\": \"[^ ]*/pmc1080/summary.txt#\"source\": \"profiles/$round/pmc_summary_1080p_b64.txt#" "$P/traffic.json"
sed -i "s#\"source\": \"[^ ]*/pmc4k/summary.txt#\"source\": \"profiles/$round/pmc_summary_4k_l5_b16.txt#" "$P/traffic_4k.json"
sed -i "s#\"source\": \"[^ ]*/pmc720/summary.txt#\"source\": \"profiles/$round/pmc_summary_720p_b1.txt#" "$P/traffic_720p.json"
sed -i "s#^[^ ]*/kt1080/runc/#(box) kt1080/runc/#; s#^[^ ]*/kt4k/runc/#(box) kt4k/runc/#; s#^[^ ]*/kt720/runc/#(box) kt720/runc/#" "$P"/sweep_busy_*.txt
ls "$P" | wc -l
