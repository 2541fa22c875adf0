"""Anatomy of the last full steps of a bench run from a rocprofv3 --kernel-trace CSV: per kernel class the launches, summed and union
busy time per step; how long 0 / 1 / 2 sweep launches were in flight; the device's idle time (no kernel at all) as a histogram of
gap lengths with the kernel that ended each gap.
CAVEAT printed with the output: under the tracer the HOST side of the ~1 700 launches per step becomes the bottleneck (a traced step
takes ~4 % longer at 1080p, ~16 % at 4K) and most long gaps sit where one stream has run dry waiting for the host; bench.py's
`device_busy_ms` is the untraced figure.
usage: python tools/step_anatomy.py <..._kernel_trace.csv> [steps=3]"""
import collections
import csv
import sys


def union(iv):
    iv = sorted(iv)
    u, (lo, hi) = 0, iv[0]
    for a, b in iv[1:]:
        if a > hi:
            u += hi - lo
            lo, hi = a, b
        elif b > hi:
            hi = b
    return u + hi - lo


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]))
    rows.sort()
    phis = [i for i, r in enumerate(rows) if r[2].startswith("k_phi_mask")]
    if len(phis) < nsteps + 1:
        sys.exit(f"need {nsteps + 1} phi launches (steps), found {len(phis)}")
    win = rows[phis[-nsteps - 1] + 1:phis[-1] + 1]
    t0, t1 = win[0][0], max(r[1] for r in win)
    busy = union([(a, b) for a, b, _ in win])
    print(f"{path}: last {nsteps} steps, {(t1 - t0) / 1e6 / nsteps:.3f} ms per step under the tracer")
    print(f"  device busy {busy / 1e6 / nsteps:.3f} ms per step, idle {(t1 - t0 - busy) / 1e6 / nsteps:.3f} ms per step "
          f"(tracer caveat: the host is the bottleneck of a traced run; see bench.py device_busy_ms for the untraced figure)")
    cls = collections.defaultdict(list)
    for a, b, n in win:
        cls[n].append((a, b))
    print("  kernel class                    launches/step   sum ms/step   union ms/step")
    for n, iv in sorted(cls.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
        print(f"  {n:32s} {len(iv) / nsteps:10.1f} {sum(b - a for a, b in iv) / 1e6 / nsteps:13.3f} {union(iv) / 1e6 / nsteps:13.3f}")
    ev = []
    for a, b, n in win:
        if n.startswith("k_blur_iter"):
            ev += [(a, 1), (b, -1)]
    ev.sort()
    cur, last, t = 0, t0, collections.Counter()
    for x, d in ev:
        t[cur] += x - last
        last, cur = x, cur + d
    t[0] += t1 - last
    print("  sweep launches in flight (ms per step): " + ", ".join(f"{k}: {v / 1e6 / nsteps:.3f}" for k, v in sorted(t.items())))
    edges = [1, 2, 4, 8, 16, 32, 64, 1e9]
    hist, tot, after = collections.Counter(), collections.Counter(), collections.defaultdict(collections.Counter)
    hi = win[0][1]
    for a, b, n in win[1:]:
        if a > hi:
            g = (a - hi) / 1e3
            e = next(e for e in edges if g <= e)
            hist[e] += 1; tot[e] += g; after[e][n] += 1
        hi = max(hi, b)
    print("  idle gaps of the whole device, by length (us): count per step, ms per step, what started after them")
    lo = 0
    for e in edges:
        if hist[e]:
            print(f"    ({lo:>2}, {e if e < 1e9 else 'inf':>3}]  {hist[e] / nsteps:7.1f}  {tot[e] / 1e3 / nsteps:7.3f}   " +
                  ", ".join(f"{k} x{v}" for k, v in after[e].most_common(3)))
        lo = e


if __name__ == "__main__":
    main()
