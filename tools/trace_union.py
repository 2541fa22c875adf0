"""Busy time of a kernel from a rocprofv3 --kernel-trace CSV: the union of its dispatches' [start, end) intervals next to their sum.
With two pairs in flight (two streams) sweep launches overlap; a bandwidth figure has to be quoted on the union, which the
kernel_stats summary (sum / calls) cannot show.
usage: python tools/trace_union.py <..._kernel_trace.csv> [kernel-name-substring ...]     (default: k_blur_iter_fast)
       --bytes N                algorithmic bytes moved by all matching dispatches together (prints GB/s on the union and on the sum)
       --bytes-per-dispatch N   the same as an average per dispatch (bench.py: roofline.alg_bytes_per_launch_avg)"""
import csv
import sys


def main():
    args = sys.argv[1:]
    nbytes = None
    if "--bytes" in args:
        i = args.index("--bytes")
        nbytes = float(args[i + 1])
        del args[i:i + 2]
    per = None
    if "--bytes-per-dispatch" in args:
        i = args.index("--bytes-per-dispatch")
        per = float(args[i + 1])
        del args[i:i + 2]
    path, pats = args[0], (args[1:] or ["k_blur_iter_fast"])
    iv = []
    for row in csv.DictReader(open(path)):
        if any(p in row["Kernel_Name"] for p in pats):
            iv.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    iv.sort()
    total = sum(b - a for a, b in iv)
    union, lo, hi = 0, None, None
    for a, b in iv:
        if hi is None or a > hi:
            if hi is not None:
                union += hi - lo
            lo, hi = a, b
        elif b > hi:
            hi = b
    if hi is not None:
        union += hi - lo
    n = len(iv)
    if per:
        nbytes = per * n
    print(f"{path}: {n} dispatches matching {pats}")
    print(f"  sum of durations   {total / 1e6:10.3f} ms   average {total / max(n, 1) / 1e3:8.2f} us")
    print(f"  union (busy time)  {union / 1e6:10.3f} ms   dispatches in flight on average {total / max(union, 1):.2f}")
    if nbytes:
        print(f"  {nbytes / 1e9:.2f} GB algorithmic: {nbytes / union:.1f} GB/s on the union, {nbytes / total:.1f} GB/s on the sum")


if __name__ == "__main__":
    main()
