"""Timeline of a rocprofv3 --kernel-trace CSV: the last N dispatches with start offset, duration and the gap to the previous end,
then the device's idle intervals as a histogram (time in which NO kernel runs, whatever the stream).
usage: python tools/trace_timeline.py <..._kernel_trace.csv> [--last N] [--from-kernel SUBSTR] [--hist]"""
import csv
import sys


def short(name):
    name = name.split("(")[0]
    for p in ("void ", "(anonymous namespace)::"):
        name = name.replace(p, "")
    return name[:44]


def main():
    args = sys.argv[1:]
    last, hist = 60, False
    if "--last" in args:
        i = args.index("--last"); last = int(args[i + 1]); del args[i:i + 2]
    if "--hist" in args:
        args.remove("--hist"); hist = True
    rows = []
    for r in csv.DictReader(open(args[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    tail = rows[-last:]
    t0 = tail[0][0]
    hi = None
    for a, b, name, q in tail:
        gap = (a - hi) / 1e3 if hi is not None else 0.0
        print(f"{(a - t0) / 1e3:9.2f} us  dur {(b - a) / 1e3:7.2f}  gap {gap:7.2f}  q{q[-3:]:>3s}  {short(name)}")
        hi = b if hi is None else max(hi, b)
    print(f"window: {(max(r[1] for r in tail) - t0) / 1e3:.2f} us, sum of durations {sum(b - a for a, b, _, _ in tail) / 1e3:.2f} us")
    if hist:
        # idle intervals of the whole device between the first and the last dispatch
        edges = [1, 2, 4, 8, 16, 32, 64, 1e9]
        cnt = [0] * len(edges); tot = [0.0] * len(edges)
        hi = rows[0][1]
        busy = rows[0][1] - rows[0][0]
        lo = rows[0][0]
        for a, b, _, _ in rows[1:]:
            if a > hi:
                g = (a - hi) / 1e3
                for k, e in enumerate(edges):
                    if g <= e:
                        cnt[k] += 1; tot[k] += g
                        break
                busy += b - a
                hi = b
            elif b > hi:
                busy += b - hi
                hi = b
        span = (hi - lo) / 1e3
        print(f"span {span / 1e3:.3f} ms, device busy {busy / 1e6:.3f} ms, idle {span / 1e3 - busy / 1e6:.3f} ms")
        prev = 0
        for e, c, t in zip(edges, cnt, tot):
            print(f"  idle gaps ({prev}, {e if e < 1e9 else 'inf'}] us: {c:7d} gaps, {t / 1e3:8.3f} ms")
            prev = e


if __name__ == "__main__":
    main()
