#!/usr/bin/env python3
"""bench.py -- frame-pairs/s of the hot path (frames -> Farneback flow -> FoE -> phi -> masks -> box) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1: this process; N > 1: one rank per GPU, started by
                                                            torch.distributed.run or by bench.py's own launcher)

A step is one pass of the whole path over one batch of synthetic 1920x1080 pairs (BASELINE config 3: batch 64 per
GPU) with the frames already resident in HBM.  Prints ONE JSON line (rank 0).  Multi-GPU: every rank runs its own
batch (weak scaling, no data-path collective) and the per-pair 32-byte result records are all-gathered over
RCCL/xGMI at the end of each step, inside the timed region, on the context's own stream (mav_allgather_results).

After the timed loop, outside the timed region, the outputs the LAST timed step left behind are checked against the
oracle (records, both masks and boxes bit for bit against the numpy chain evaluated on that step's own flow; that flow
against the CPU Farneback): `verified_pairs`.  A mismatch makes the process exit non-zero.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "mav-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _cgroup_cores():
    """Host cores the cgroup's CPU-time quota amounts to (None: no quota)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else max(1, int(float(quota) / float(period) + 0.5))
    except (OSError, ValueError):
        return None


# numpy's BLAS starts one thread per core it can SEE (64 on the 256-core GPU hosts) and its idle threads spin for tens of ms after a
# matrix product; inside a cgroup with a CPU-time quota (16 cores per 100 ms here) the spinners burn the quota and the kernel freezes
# every thread of the process until the period ends: 25 - 80 ms stalls in whatever runs next (profiles/r06/stall_bisect.txt).  Half
# the granted cores at most, set before numpy is imported.
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, str(max(1, (_cgroup_cores() or (os.cpu_count() or 2)) // 2)))

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
ITER_BYTES_UPDATE = 88           # SURVEY 8d's model, per pixel per sweep: read M 20 + R0 20 + R1 20, write M' 20 + flow 8
ITER_BYTES_MOVED = 80            # what the kernel has to move: the flow of an updating sweep is consumed inside the kernel (store_flow = 0)
ITER_BYTES_LAST = 28             # last sweep of a layer: read M 20, write flow 8
PROFILE_ROUND = "r06"


def b_alg_per_pair(layers, W, H, iters):
    """Algorithmic bytes of one pair, SURVEY 8(d): per layer blur+resize 2(P0+4n), polyexp 2(4n+20n), initial M 60n
    (+8 n_coarser), sweeps (I-1)*88n + 28n; then phi+threshold 9 P0 and box 1 P0."""
    P0 = W * H
    tot = 0
    for k, (w, h) in enumerate(layers):
        n = w * h
        tot += 2 * (P0 + 4 * n) + 2 * (4 * n + 20 * n) + 60 * n + (iters - 1) * ITER_BYTES_UPDATE * n + ITER_BYTES_LAST * n
        if k + 1 < len(layers):
            tot += 8 * layers[k + 1][0] * layers[k + 1][1]
    return tot + 10 * P0


def source_hash(schedule=None):
    """Identity of what was measured: sha256 over the library's sources (the GPU box has no .git) and, when given, the schedule the
    call takes (mav_schedule_info: every option in effect plus the per-layer plan) -- the library reads no environment variable, so
    sources + options determine the launches."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mav-detection_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    if schedule is not None:
        h.update(json.dumps(schedule, sort_keys=True).encode())
    return h.hexdigest()[:16]


def usable_cores():
    """Host cores this process may actually run on: the affinity mask, capped by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = _cgroup_cores()
    return n if q is None else max(1, min(n, q))


def pin_rank_to_its_cores(local_rank: int, local_world: int):
    """N ranks share the node's host cores (and, on this pool, one cgroup quota of 16 cores): each rank's threads -- the enqueueing
    thread, the staging threads, the runtime's helpers -- are confined to a core set of its own, usable cores // ranks wide, BEFORE
    anything touches the GPU (threads started later inherit the mask).  Returns the number of cores this rank got; a single rank
    keeps every usable core (its CPU-baseline leg wants them)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return usable_cores()
    cores = sorted(os.sched_getaffinity(0))
    per = max(1, min(len(cores), usable_cores()) // local_world)
    mine = cores[local_rank * per:(local_rank + 1) * per] or cores[-per:]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return usable_cores()
    return len(mine)


def cpu_flow_fn():
    """cv2.calcOpticalFlowFarneback when OpenCV is importable on the node (kind "reference"), else the oracle's C restatement
    (kind "port", labelled as such)."""
    try:
        import cv2
        cv2.setNumThreads(1)
        return (lambda a, b: cv2.calcOpticalFlowFarneback(a, b, None, 0.4, 1, 12, 10, 8, 1.2, 0)), "reference", \
            f"cv2 {cv2.__version__} calcOpticalFlowFarneback (1 thread per pair)"
    except ImportError:
        from oracle import fb_oracle
        return fb_oracle.load().calc, "port", "oracle/farneback_oracle.c (restatement, not OpenCV: cv2 is not importable on this node)"


def cpu_baseline(prev, nxt, samples, n_sample, levels):
    """The CPU path on a bounded sample of the same workload, timed on this node's host cores: once on ONE core, and once
    frame-parallel on every core this process may use.  The numpy FoE chain follows the flow in both."""
    from oracle import foe_oracle
    flow_fn, kind, label = cpu_flow_fn()
    if levels != 1:
        from oracle import fb_oracle
        orc, par = fb_oracle.load(), fb_oracle.default_params(levels)
        flow_fn, kind, label = (lambda a, b: orc.calc(a, b, par)), "port", f"oracle/farneback_oracle.c, levels={levels} (restatement, not OpenCV)"
    B = prev.shape[0]
    t0 = time.perf_counter()
    for b in range(n_sample):
        foe_oracle.run_chain(flow_fn(prev[b % B], nxt[b % B]), samples[b % B])
    dt = time.perf_counter() - t0
    cores = usable_cores()
    out = {"value": n_sample / dt, "unit": "frame-pairs/s", "cores": 1, "kind": kind,
           "sample": f"{n_sample} of the benchmark's {prev.shape[2]}x{prev.shape[1]} pairs, {label} + numpy FoE chain, {dt:.1f} s on one core; "
                     f"host has {os.cpu_count()} cores, {cores} usable by this process"}
    if cores > 1:
        # frame-parallel over all usable cores (threads: the C flow call and numpy's inner loops release the GIL); every
        # thread gets the same number of pairs, at least as many pairs as cores
        from concurrent.futures import ThreadPoolExecutor
        per = max(1, min(8, int(10.0 / max(dt / n_sample, 1e-3))))       # about 10 s of work per core
        n_all = cores * per

        def one(i):
            foe_oracle.run_chain(flow_fn(prev[i % B], nxt[i % B]), samples[i % B])
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(one, range(n_all)))
        dtp = time.perf_counter() - t0
        out["all_cores"] = {"value": n_all / dtp, "unit": "frame-pairs/s", "cores": cores, "host_cores": os.cpu_count(),
                            "pairs": n_all, "seconds": round(dtp, 2), "kind": kind}
    return out


def verify_last_step(ctx, prev, nxt, samples, res, mf_buf, md_buf, pairs, levels):
    """Outside the timed region: the records / masks / flow the last timed step left on the device, against the oracle."""
    import numpy as np
    from oracle import foe_oracle, tolerances
    flow_fn, kind, _ = cpu_flow_fn()
    if levels != 1:
        from oracle import fb_oracle
        orc, par = fb_oracle.load(), fb_oracle.default_params(levels)
        flow_fn, kind = (lambda a, b: orc.calc(a, b, par)), "port"
    H, W = prev.shape[1:]
    ok, bad, epes = [], [], []
    for b in pairs:
        flow = ctx.last_flow(b)
        chain = foe_oracle.run_chain(flow, samples[b])
        mf = np.empty((H, W), np.uint8)
        md = np.empty((H, W), np.uint8)
        ctx.lib.mav_memcpy_d2h(ctx.h, mf.ctypes.data, mf_buf.ptr + b * W * H, W * H)
        ctx.lib.mav_memcpy_d2h(ctx.h, md.ctypes.data, md_buf.ptr + b * W * H, W * H)
        e = tolerances.epe(flow, flow_fn(prev[b], nxt[b]))
        epes.append(e.ravel())
        good = (tuple(res[b]["foe"]) == tuple(chain["foe"]) and tuple(res[b]["box"]) == tuple(chain["box"])
                and np.array_equal(mf.view(np.bool_), chain["fixed"]) and np.array_equal(md.view(np.bool_), chain["total"])
                and tolerances.flow_epe_ok(e))
        (ok if good else bad).append(int(b))
    e = np.concatenate(epes)
    return {"verified_pairs": ok, "failed_pairs": bad,
            "checked": "records (FoE, box) and both masks of the last timed step bit-exact vs the numpy chain on that step's own flow; "
                       "flow EPE vs the CPU Farneback within " + tolerances.FLOW_GATE_TEXT,
            "flow_epe_px": {"mean": float(e.mean()), "p99.9": float(np.percentile(e, 99.9)), "max": float(e.max()),
                            "against": "cv2" if kind == "reference" else "oracle restatement (cv2 absent)", "pairs": len(pairs)}}


def equal_plain_schedule(ctx, run_batch, d_res, d_mf, d_md, B, W, H):
    """EVERY pair of the batch the timed loop just computed against the same batch re-run in the plain schedule (one stream, one pair
    after the other: the form the parity tests check against the oracle): records, both masks and the whole flow must be identical.
    Returns (all equal, pairs with identical flow, differing pairs)."""
    import zlib
    import numpy as np
    from mavflow import _lib
    rec = _lib.RESULT_DTYPE.itemsize

    def digest():
        return (d_res.download(np.uint8, (B * rec,)).tobytes(), zlib.crc32(d_mf.download(np.uint8, (B * W * H,))),
                zlib.crc32(d_md.download(np.uint8, (B * W * H,))), [zlib.crc32(ctx.last_flow(b)) for b in range(B)])
    timed = digest()
    plain_opts = {"pairs_in_flight": 1, "deep_batch": 0, "coarse_bands": 0, "band_phase": 0}     # one stream, every layer per group
    saved = {k: ctx.get_option(k) for k in plain_opts}
    for k, v in plain_opts.items():
        ctx.set_option(k, v)
    run_batch()
    ctx.sync()
    plain = digest()
    for k, v in saved.items():
        ctx.set_option(k, v)
    same = [b for b in range(B) if timed[3][b] == plain[3][b]]
    ok = bool(timed[:3] == plain[:3] and len(same) == B)
    differing = sorted(set(range(B)) - set(same)) if not ok else []
    return ok, len(same), (differing if differing or ok else [-1])


def host_enqueue_ms(ctx, step, reps=7):
    """Host time of ENQUEUEING one step: the wall time of the call(s) of one step issued into an idle queue (sync before, clock stopped
    when the last call returns, before any sync), median of `reps`.  What one host thread per rank has to sustain."""
    ts = []
    for _ in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        step()
        ts.append(1e3 * (time.perf_counter() - t0))
        ctx.sync()
    return sorted(ts)[len(ts) // 2]


def traffic_record(want):
    """HBM-side request bytes of the sweep launches of one step from the committed PMC record that matches this build, shape, batch and
    schedule (profiles/<round>/traffic*.json, written by tools/pmc_passes.sh + tools/make_traffic.py): (bytes per launch or None, source)."""
    pdir = os.path.join(ROOT, "profiles", PROFILE_ROUND)
    why = []
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not (name.startswith("traffic") and name.endswith(".json")):
            continue
        rec_t = json.load(open(os.path.join(pdir, name)))
        have = {k: rec_t.get(k) for k in want}
        if have == want:
            return (int(rec_t["hbm_bytes_sweeps_per_step"] / max(want["launches"], 1)),
                    f"profiles/{PROFILE_ROUND}/{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build, shape, batch and "
                    f"schedule (memory-side request bytes: Infinity-Cache hits included, so not DRAM bytes); not measured by this run")
        why.append(f"{name}: " + ", ".join(f"{k} {have[k]} != {want[k]}" for k in want if have[k] != want[k]))
    return None, ("no committed PMC record matches this build / shape (" + "; ".join(why) + "): null" if why else "no PMC record committed: null")


def trace_record(W, H, B, levels):
    """The same union from the committed rocprofv3 kernel trace of this configuration (profiles/<round>/sweep_busy_*.txt, written by
    tools/profile_round.sh on a traced run of the headline loop): (GB/s on the union, traced ms per step) or (None, None).  A traced
    step is slower than an untraced one -- the tracer's cost per dispatch falls on ~1 600 launches per step -- so this figure is the
    lower of the two; the bench line carries both."""
    import re
    name = {(1920, 1080, 64, 1): "1080p_b64", (3840, 2160, 16, 5): "4k_l5_b16", (1280, 720, 1, 1): "720p_b1"}.get((W, H, B, levels))
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, f"sweep_busy_{name}.txt") if name else None
    if not path or not os.path.exists(path):
        return None, None
    txt = open(path).read()
    m = re.search(r"([0-9.]+) GB/s on the union", txt)
    t = re.search(r"traced step\s+([0-9.]+) ms", txt)
    return (float(m.group(1)) if m else None), (float(t.group(1)) if t else None)


def sweep_roofline(ctx, run_batch, B, layers, W, H, levels, schedule, step_ms, ceil=None):
    """Roofline of the dominant kernel (the sweeps) for the workload run_batch() enqueues, measured live with HIP events in two extra
    passes outside any timed region: (1) events around every launch -> per-class sums, launch counts, avg launch duration (what
    rocprofv3 --kernel-trace --stats reports per dispatch); (2) events around every RUN of launches of one class on a stream (a tenth
    of the events, so the two streams overlap as in the timed loop) -> the union of the intervals in which a sweep launch runs."""
    ctx.profile_enable(True)
    run_batch()
    prof = ctx.profile_get()
    busy_per_launch_events = ctx.profile_busy("blur_iter", "blur_iter_coarse")
    ctx.profile_enable(False)
    ctx.profile_enable(2)
    run_batch()
    busy_ms = ctx.profile_busy("blur_iter", "blur_iter_coarse")
    busy_all_ms = ctx.profile_busy(*prof.keys())        # time during which ANY kernel of the step was running (both streams)
    ctx.profile_enable(False)
    sum_ms = prof["blur_iter"][0] + prof.get("blur_iter_coarse", (0.0, 0))[0]
    launches = prof["blur_iter"][1] + prof.get("blur_iter_coarse", (0, 0))[1]
    iters = ctx.fb.iterations
    npx_step = B * sum(w * h for (w, h) in layers)
    bytes_moved = npx_step * ((iters - 1) * ITER_BYTES_MOVED + ITER_BYTES_LAST)
    bytes_survey = npx_step * ((iters - 1) * ITER_BYTES_UPDATE + ITER_BYTES_LAST)
    # Two pairs are in flight on two streams (the library's default schedule): sweep launches overlap pairwise, so the SUM of their
    # durations exceeds the wall time.  The rate is quoted on the time during which the kernel was running at all -- the union of
    # the launches' intervals; avg_launch_ms stays the per-launch figure rocprofv3 reports (tools/trace_union.py computes the same
    # union from a rocprofv3 kernel trace).
    achieved = bytes_moved / (busy_ms * 1e-3) / 1e9
    achieved_survey = bytes_survey / (busy_ms * 1e-3) / 1e9
    if ceil is None:
        ceil = measured_ceilings(ctx)
    # HBM bytes from the PMC counters: collected by tools/pmc_passes.sh in separate rocprofv3 --pmc runs of THIS workload, corrected
    # as the microarchitecture guide prescribes, committed under profiles/ with the hash of the sources and schedule they were
    # measured on.  Per launch, like `achieved`.  null unless the record matches this build, shape, batch and schedule.
    traffic, traffic_source = traffic_record({"source_hash": source_hash(schedule), "width": W, "height": H, "batch": B, "levels": levels,
                                              "launches": launches})
    from_trace, traced_step_ms = trace_record(W, H, B, levels)
    return {"bound": "hbm", "served_by": "infinity_cache", "kernel": "k_blur_iter_fast (all sweep launches of a step)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "achieved_from_trace": from_trace, "frac_from_trace": None if from_trace is None else round(from_trace / HBM_PEAK_GBS, 4),
            "traced_step_ms": traced_step_ms, "untraced_step_ms": round(step_ms, 3),
            "achieved_is": "effective algorithmic bandwidth = bytes the sweeps have to move (80 B/px per updating sweep: M, R0, R1 in, M' out -- "
                           "the flow of all but a layer's last sweep never leaves the kernel; 28 B/px for the last) / time during which at least one "
                           "sweep launch runs.  The schedule keeps every pair's band of M / R0 / R1 inside the 256 MB Infinity Cache between "
                           "sweeps, so most of these bytes are served by the cache: this is NOT a DRAM rate, and `frac` compares it with the "
                           "8 TB/s HBM spec only because that is the contract's denominator",
            "achieved_survey_88B_model": round(achieved_survey, 1), "frac_survey_88B_model": round(achieved_survey / HBM_PEAK_GBS, 4),
            "measured_ceiling": ceil,
            "frac_of_measured_ceiling": round(achieved / max(ceil["infinity_cache_GBs"], 1e-9), 4),
            "traffic": traffic, "traffic_source": traffic_source,
            "avg_launch_ms": round(sum_ms / max(launches, 1), 4), "launches_per_step": launches,
            "kernel_launches_per_step_all_classes": int(sum(v[1] for v in prof.values())),
            "kernel_busy_ms": round(busy_ms, 3), "kernel_busy_ms_with_per_launch_events": round(busy_per_launch_events, 3),
            "sum_of_launch_ms": round(sum_ms, 3),
            "launches_in_flight": round(sum_ms / max(busy_per_launch_events, 1e-9), 2),
            "note": "two pairs are in flight on two streams: sweep launches overlap, `achieved` = bytes of all sweep launches / kernel_busy_ms "
                    "(union of the intervals in which the kernel runs, HIP events around every run of launches on each stream); avg_launch_ms / "
                    "sum_of_launch_ms come from a pass with events around every launch, whose sum exceeds the step by design; the same union from a "
                    f"rocprofv3 kernel trace: profiles/{PROFILE_ROUND}/sweep_busy_*.txt",
            "alg_bytes_per_launch_avg": int(bytes_moved / max(launches, 1)),
            "alg_bytes_per_launch_avg_survey_88B_model": int(bytes_survey / max(launches, 1)),
            "kernel_share_of_step": round(busy_ms / step_ms, 3),
            "device_busy_ms": round(busy_all_ms, 3),
            "device_busy_note": "union of ALL kernel classes' intervals in the one-step pass with HIP events around runs of launches (no tracer: "
                                "under rocprofv3 the host side of the launches becomes the bottleneck and idle gaps appear that the untraced "
                                "run does not have); step time minus this = time in which nothing runs on the device",
            "all_kernels_ms": {k: round(v[0], 3) for k, v in prof.items()}}


def run_config_leg(name, W, H, B, levels, calls, pairs_to_check, verify=True, ceil=None):
    """One more BASELINE configuration through mav_process_batch_dev in a context of its own (never `value`): `calls` back-to-back
    calls with the frames resident, HIP-event time on the context's stream and wall time; then the last call's outputs of
    `pairs_to_check` against the oracle.  Returns the record for the bench line's "configs" object."""
    import numpy as np
    from mavflow import _lib, synth
    ctx = _lib.Context(W, H, B, _lib.fb_defaults(levels=levels))
    layers = [ctx.layer_dims(k)[:2] for k in range(ctx.num_layers())]
    prev, nxt = synth.make_batch(W, H, B, distinct=min(B, 4))
    samples = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
    d_prev, d_next, d_smp = ctx.alloc(prev.nbytes).upload(prev), ctx.alloc(nxt.nbytes).upload(nxt), ctx.alloc(samples.nbytes).upload(samples)
    d_res, d_mf, d_md = ctx.alloc(B * _lib.RESULT_DTYPE.itemsize), ctx.alloc(B * W * H), ctx.alloc(B * W * H)

    def call():
        ctx.process_batch_dev(d_prev.ptr, d_next.ptr, d_smp.ptr, B, d_res.ptr, mf_ptr=d_mf.ptr, md_ptr=d_md.ptr)

    for _ in range(max(10, calls // 2)):               # (the first calls after an idle phase run slower: clocks ramp up)
        call()
    ctx.sync()
    # three repetitions of `calls` back-to-back calls; the median repetition is reported (a launch-latency-bound chain of ~27 small
    # kernels per call is sensitive to the clock state the preceding legs left the GPU in), all three are listed
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        ctx.timer_start()
        for _ in range(calls):
            call()
        ev = ctx.timer_stop()
        ctx.sync()
        reps.append((ev, 1e3 * (time.perf_counter() - t0)))
    ev_ms, wall_ms = sorted(reps)[1]
    balg = b_alg_per_pair(layers, W, H, ctx.fb.iterations)
    per_pair_ms = ev_ms / calls / B
    out = {"workload": f"{W}x{H}, batch={B}, levels={levels} ({len(layers)} pyramid layers), 3 x {calls} back-to-back mav_process_batch_dev calls "
                       f"(median repetition reported), frames resident",
           "ms_per_call_hip_events": round(ev_ms / calls, 4), "ms_per_call_wall": round(wall_ms / calls, 4),
           "ms_per_call_hip_events_all_reps": [round(e / calls, 4) for e, _ in reps],
           "ms_per_pair": round(per_pair_ms, 4), "pairs_per_s": round(B * calls / (wall_ms * 1e-3), 2),
           "alg_bytes_per_pair": balg, "frac": round(balg / (per_pair_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
           "frac_is": "whole-pipeline algorithmic bytes (SURVEY 8d) / HIP-event time / 8 TB/s",
           "schedule": ctx.schedule_info(B)}
    if verify:
        res = d_res.download(_lib.RESULT_DTYPE, (B,))
        v = verify_last_step(ctx, prev, nxt, samples, res, d_mf, d_md, pairs_to_check, levels)
        out.update({"verified_pairs": v["verified_pairs"], "failed_pairs": v["failed_pairs"], "flow_epe_px": v["flow_epe_px"]})
        ok, n_same, differing = equal_plain_schedule(ctx, call, d_res, d_mf, d_md, B, W, H)
        out["all_pairs_equal_plain_schedule"] = ok
        out["pairs_with_identical_flow_in_plain_schedule"] = n_same
        if not ok:
            out["failed_pairs"] = sorted(set(out["failed_pairs"]) | set(differing))
    # the dominant kernel of THIS configuration, untraced (same two passes as the headline's roofline block), and the host side
    out["host_enqueue_ms_per_call"] = round(host_enqueue_ms(ctx, call), 4)
    out["roofline"] = sweep_roofline(ctx, call, B, layers, W, H, levels, out["schedule"], ev_ms / calls, ceil)
    out["host_enqueue_share_of_call"] = round(out["host_enqueue_ms_per_call"] / (ev_ms / calls), 3)
    if B == 1:
        # BASELINE config 2 is a per-frame detector: the distribution of ONE call's latency (enqueue + wait), 1 200 calls
        lat = []
        for _ in range(1200):
            t0 = time.perf_counter(); call(); ctx.sync(); lat.append(1e3 * (time.perf_counter() - t0))
        lat = np.sort(np.asarray(lat))
        out["latency_calls"], out["median_ms"], out["p99_ms"], out["max_ms"] = len(lat), round(float(np.median(lat)), 4), round(float(lat[int(0.99 * len(lat))]), 4), round(float(lat[-1]), 4)
        out["lanes"] = lanes_leg(ctx, W, H, levels, (d_prev, d_next, d_smp), calls, balg)
        if not out["lanes"]["records_identical_across_contexts"]:
            out["failed_pairs"] = out.get("failed_pairs", []) + [-1]
    ctx.close()
    return out


def lanes_leg(ctx0, W, H, levels, inputs, calls, balg):
    """A STREAM of one-pair calls given to 2 and 3 contexts in turn (each context owns its streams and workspace: consecutive calls on
    different contexts are independent chains of launches that the GPU interleaves -- mavflow/pipeline.py `auto_lanes`, which the
    one-frame loop of the API uses).  Throughput of the stream, wall clock over all contexts; the latency of a call stays what
    `ms_per_call_hip_events` says."""
    import numpy as np
    from mavflow import _lib
    d_prev, d_next, d_smp = inputs
    n0 = W * H
    extra = []
    for _ in range(2):
        c = _lib.Context(W, H, 1, _lib.fb_defaults(levels=levels))
        b = [c.alloc(n0), c.alloc(n0), c.alloc(d_smp.nbytes), c.alloc(_lib.RESULT_DTYPE.itemsize), c.alloc(n0), c.alloc(n0)]
        b[0].upload(d_prev.download(np.uint8, (n0,))); b[1].upload(d_next.download(np.uint8, (n0,))); b[2].upload(d_smp.download(np.uint8, (d_smp.nbytes,)))
        extra.append((c, b))
    c0 = (ctx0, [d_prev, d_next, d_smp, ctx0.alloc(_lib.RESULT_DTYPE.itemsize), ctx0.alloc(n0), ctx0.alloc(n0)])
    out = {"is": "wall-clock ms per pair of a stream of one-pair mav_process_batch_dev calls taken by N contexts in turn (throughput; a call's "
                 "latency is unchanged); same frames, same results in every context"}
    recs = []
    for n in (1, 2, 3):
        lanes = ([c0] + extra)[:n]
        for k in range(30):
            c, b = lanes[k % n]
            c.process_batch_dev(b[0].ptr, b[1].ptr, b[2].ptr, 1, b[3].ptr, mf_ptr=b[4].ptr, md_ptr=b[5].ptr)
        for c, _ in lanes:
            c.sync()
        reps = []
        for _ in range(3):
            t0 = time.perf_counter()
            for k in range(calls):
                c, b = lanes[k % n]
                c.process_batch_dev(b[0].ptr, b[1].ptr, b[2].ptr, 1, b[3].ptr, mf_ptr=b[4].ptr, md_ptr=b[5].ptr)
            for c, _ in lanes:
                c.sync()
            reps.append(1e3 * (time.perf_counter() - t0) / calls)
        ms = sorted(reps)[1]
        out[f"{n}_context" + ("s" if n > 1 else "")] = {"ms_per_pair": round(ms, 4), "frac": round(balg / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4)}
        recs.append(b"".join(b[3].download(np.uint8, (_lib.RESULT_DTYPE.itemsize,)).tobytes() for _, b in lanes))
    out["records_identical_across_contexts"] = bool(all(len(set(r[i:i + 32] for i in range(0, len(r), 32))) == 1 for r in recs))
    for c, b in extra:
        c.close()
    return out


def api_loop_leg(W=1920, H=1080, batch=64, n_batches=12, n_unbatched=288, staged=False, only_batched=False):
    """The door north_star says users come through: the reference-shaped loops of mavflow.processor on a pre-generated synthetic
    dataset -- host numpy frames in (pageable, one array per frame, as Dataset hands them out), filled FrameResults out.  Never
    `value`.  Each loop runs once to warm its context (workspace allocation, first launches) and is timed on its second run over the
    same Processor.  Frame synthesis is excluded (8 distinct pairs, generated before any clock starts)."""
    import logging
    import numpy as np
    from mavflow.processor import Processor, SyntheticDataset
    from mavflow.run_config import RunConfig

    def make(N, use_fb=True):
        ds = SyntheticDataset(W, H, N, use_farneback=use_fb, distinct=8, dangle=(0.004, -0.002, 0.001))
        for i in range(8):
            ds._pair(i); ds.get_gt_of(i)
        for _ in range(N):
            ds.get_frame()
        ds.get_segmentation(0); ds.get_sky_segmentation(0); ds.get_depth(0)
        cfg = RunConfig(logging.getLogger("bench"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING")
        return Processor(cfg), ds

    def timed(p, ds, fn, N):
        ds.N = min(N, 3 * batch + 1)
        fn()                                               # warm: context, workspace, all three pipeline slots
        p.frame_index = 0; p.detection_results = {}; p.config.results = {}
        # (the warm run's last flow / masks, which the loop leaves behind as attributes: a fresh run holds none -- kept, the timed run's
        # first two batches would bring them to the host before their slots are re-used: 8 ms of page-locked allocation each)
        p.flow_uv = p.estimate_fixed = p.total_mask = None
        ds.N = N
        np.random.seed(7)
        t0 = time.perf_counter()
        res = fn()
        dt = time.perf_counter() - t0
        assert len(res) == N - 1 and all(r.tpr is not None for r in res.values())
        return dt, res

    out = {"workload": f"{W}x{H} SyntheticDataset (8 distinct pairs cycled, pageable numpy frames), Processor loops of mavflow.processor; "
                       f"second run of each loop timed, frame synthesis excluded"}
    N = batch * n_batches + 1
    p, ds = make(N)
    dt, res_b = timed(p, ds, lambda: p.run_detection_batched(batch=batch), N)
    out["run_detection_batched"] = {"batch": batch, "pairs": N - 1, "pairs_per_s": round((N - 1) / dt, 1), "ms_per_pair": round(1e3 * dt / (N - 1), 4)}
    p.release()
    if only_batched:
        return out
    N1 = n_unbatched + 1
    p, ds = make(N1)
    dt, res_1 = timed(p, ds, p.run_detection, N1)
    from mavflow import pipeline
    out["run_detection"] = {"flow_seam": "Farneback on the GPU (DeviceArray)", "frames": N1 - 1, "ms_per_frame": round(1e3 * dt / (N1 - 1), 4),
                            "pairs_per_s": round((N1 - 1) / dt, 1), "lanes": pipeline.auto_lanes(W, H, 1)}
    same = all(vars(res_1[i]) == vars(res_b[i]) for i in range(min(N1, N) - 1))
    out["batched_and_unbatched_results_identical"] = bool(same)
    p.release()
    p, ds = make(N1, use_fb=False)
    dt, _ = timed(p, ds, p.run_detection, N1)
    out["run_detection_host_flow"] = {"flow_seam": "float32 host array per frame (what a .flo file gives)", "frames": N1 - 1,
                                      "ms_per_frame": round(1e3 * dt / (N1 - 1), 4)}
    p.release()
    if (W, H) == (1920, 1080):
        # BASELINE config 2's shape through the same door: the one-frame loop at 1280x720 (three lanes)
        W2, H2 = 1280, 720
        ds = SyntheticDataset(W2, H2, N1, use_farneback=True, distinct=8, dangle=(0.004, -0.002, 0.001))
        for i in range(8):
            ds._pair(i); ds.get_gt_of(i)
        for _ in range(N1):
            ds.get_frame()
        p = Processor(RunConfig(logging.getLogger("bench"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
        dt, _ = timed(p, ds, p.run_detection, N1)
        out["run_detection_1280x720"] = {"flow_seam": "Farneback on the GPU (DeviceArray)", "frames": N1 - 1, "ms_per_frame": round(1e3 * dt / (N1 - 1), 4),
                                         "lanes": pipeline.auto_lanes(W2, H2, 1)}
        p.release()
    if staged:
        Ns = 13
        p, ds = make(Ns)
        dt, _ = timed(p, ds, p.run_detection_staged, Ns)
        out["run_detection_staged"] = {"frames": Ns - 1, "ms_per_frame": round(1e3 * dt / (Ns - 1), 4)}
        p.release()
    return out


def measured_ceilings(ctx):
    """What this GPU delivers to a plain streaming kernel with the sweeps' 3 reads : 1 write mix (float4 per thread, grid-stride),
    measured now, in this process (mav_membw_probe): once with a footprint the 256 MB Infinity Cache holds, once far beyond it."""
    cache = ctx.membw_probe(32 << 20, 40)          # 4 x 32 MB = 128 MB footprint
    hbm = ctx.membw_probe(1 << 30, 4)              # 4 x 1 GB
    return {"kernel": "3 reads : 1 write float4 streaming (mav_membw_probe)", "infinity_cache_GBs": round(cache, 1), "hbm_GBs": round(hbm, 1),
            "footprints_MB": [128, 4096]}


class RecordExchange:
    """How the ranks of an N > 1 run meet and what moves the per-pair records between them (ONE all-gather of 32-byte
    records per batch, nothing else crosses GPUs).

      socket (default)  mavflow.rendezvous: the 128-byte ncclUniqueId, the barriers and the max-over-ranks time travel over a localhost
                        socket (Python stdlib); the records move through the LIBRARY's communicator on the context's stream
                        (mav_allgather_results = ncclAllGather), RCCL and the HIP runtime being the ones under /opt/rocm that
                        libmavflow was built against.  No torch in the process.  If the communicator does not come up on EVERY rank
                        (agreed through the store), connect() returns False and bench.py re-runs itself on the torch path.
      torch             torch.distributed (nccl = the RCCL bundled with torch, which the library then shares; falls back to torch's own
                        all_gather when the library's communicator cannot be opened).
      --rehearse-on-one-gpu: every rank on device 0; RCCL refuses that, so the records move on the host (through the store, or gloo)."""

    COMM_TIMEOUT_S = 240.0

    def __init__(self, args, rank, world, local_rank):
        self.mode, self.rehearse = args.rendezvous, args.rehearse_on_one_gpu
        self.simulate_failure = args.simulate_socket_failure
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.rdzv = self.dist = self.torch = self.mdist = self.comm = None
        self.t_local = self.t_all = self.host_all = None
        self.why_not_library = ""
        self.finished = False

    def init_torch(self):
        import torch
        from mavflow import dist as mdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        self.torch, self.mdist = torch, mdist
        self.dist, self.rank, self.world, _ = mdist.init_process_group("gloo" if self.rehearse else "nccl")
        self.tdev = "cpu" if self.rehearse else "cuda"

    def _open_library_comm(self, uid):
        """mav_comm_init + one all-gather, in a thread the caller can give up on (RCCL's bootstrap blocks for ever when a peer is missing)."""
        import threading
        box = {}

        def work():
            try:
                comm = self.ctx.comm_init(uid, self.rank, self.world)
                self.ctx.allgather(comm, self.d_res.ptr, self.nbytes, self.d_all.ptr)
                self.ctx.sync()
                box["comm"] = comm
            except Exception as e:                         # noqa: BLE001 -- reported to the caller
                box["err"] = e
        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(self.COMM_TIMEOUT_S)
        if th.is_alive():
            raise RuntimeError(f"RCCL communicator did not come up within {self.COMM_TIMEOUT_S:.0f} s")
        if "err" in box:
            raise box["err"]
        return box["comm"]

    def connect(self, ctx, d_res, nbytes) -> bool:
        import numpy as np
        self.ctx, self.d_res, self.nbytes = ctx, d_res, nbytes
        self.d_all = ctx.alloc(self.world * nbytes)
        if self.mode == "torch":
            torch = self.torch
            try:
                if self.rehearse:
                    raise RuntimeError("rehearsal on one GPU: RCCL cannot hold two ranks on one device")
                uid = torch.zeros(128, dtype=torch.uint8)
                if self.rank == 0:
                    uid = torch.from_numpy(ctx.comm_unique_id().copy())
                uid = uid.cuda()
                self.dist.broadcast(uid, src=0)
                self.comm = self._open_library_comm(uid.cpu().numpy())
            except Exception as e:                         # noqa: BLE001 -- any failure here must not lose the measurement
                self.comm, self.why_not_library = None, str(e)
                self.t_local = torch.empty(nbytes, dtype=torch.uint8, device=self.tdev)
                self.t_all = torch.empty(self.world * nbytes, dtype=torch.uint8, device=self.tdev)
            return True
        from mavflow import rendezvous
        ok, why = True, ""
        try:
            self.rdzv = rendezvous.from_env()
            uid = self.rdzv.broadcast("uid", ctx.comm_unique_id().tobytes() if self.rank == 0 else None)
            if len(uid) != 128:
                raise RuntimeError(f"ncclUniqueId of {len(uid)} bytes")
            if not self.rehearse:
                self.comm = self._open_library_comm(np.frombuffer(uid, np.uint8))
            if self.simulate_failure and self.rank == self.world - 1:
                raise RuntimeError("simulated failure (--simulate-socket-failure)")
        except Exception as e:                             # noqa: BLE001
            ok, why = False, f"{type(e).__name__}: {e}"
        all_ok = False
        if self.rdzv is not None:
            try:
                # (a peer may sit in RCCL's bootstrap for the whole communicator time-out before it can answer)
                flags = self.rdzv.allgather("up", b"1" if ok else why.encode("utf-8", "replace")[:200], timeout=self.COMM_TIMEOUT_S + 60.0)
                all_ok = all(f == b"1" for f in flags)
                if not all_ok and self.rank == 0:
                    print(f"[bench] socket rendezvous did not come up on every rank: {[f.decode('utf-8', 'replace') for f in flags]}", file=sys.stderr, flush=True)
            except Exception as e:                         # noqa: BLE001
                why = why or str(e)
        if not all_ok:
            if self.rank == 0:
                print(f"[bench] falling back to --rendezvous torch ({why or 'a peer failed'})", file=sys.stderr, flush=True)
            try:
                if self.rdzv is not None:
                    self.rdzv.close()
            except Exception:                              # noqa: BLE001
                pass
            self.d_all.free()
            return False
        return True

    def describe(self) -> str:
        if self.comm is not None:
            how = "mavflow.rendezvous (localhost socket, no torch; RCCL + HIP runtime from /opt/rocm)" if self.mode == "socket" else "torch.distributed"
            return f"mav_allgather_results (RCCL ncclAllGather on the context's stream); ncclUniqueId over {how}"
        if self.mode == "socket":
            return "rendezvous store all-gather of host records (one-GPU rehearsal: ids only, RCCL cannot hold two ranks on one device)"
        return f"torch.distributed.all_gather_into_tensor (library communicator unavailable: {self.why_not_library})"

    def exchange(self):
        import numpy as np
        if self.finished:                                  # rank 0's post-loop legs re-run the step: the ranks have parted, nothing to exchange
            return
        if self.comm is not None:
            self.ctx.allgather(self.comm, self.d_res.ptr, self.nbytes, self.d_all.ptr)
        elif self.mode == "socket":
            self.ctx.sync()
            self.host_all = self.rdzv.allgather("rec", self.d_res.download(np.uint8, (self.nbytes,)).tobytes())
        else:
            torch = self.torch
            self.ctx.sync()                                # (torch fallback) records -> torch tensor -> all_gather
            if self.tdev == "cuda":
                torch.cuda.synchronize()
            self.t_local.copy_(torch.from_numpy(self.d_res.download(np.uint8, (self.nbytes,))))
            self.mdist.allgather_records(self.dist, self.t_local, self.t_all)
            if self.tdev == "cuda":
                torch.cuda.synchronize()

    def barrier(self):
        if self.mode == "socket":
            self.rdzv.barrier("b")
        else:
            if self.tdev == "cuda":
                self.torch.cuda.synchronize()
            self.dist.barrier()
            if self.tdev == "cuda":
                self.torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if self.mode == "socket":
            return self.rdzv.allreduce_max("t", x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gathered(self):
        import numpy as np
        if self.comm is not None:
            return self.d_all.download(np.uint8, (self.world * self.nbytes,))
        if self.mode == "socket":
            return np.frombuffer(b"".join(self.host_all), np.uint8)
        return self.t_all.cpu().numpy()

    def comm_ranks(self):
        return self.ctx.comm_count(self.comm) if self.comm is not None else None

    def finish(self):
        self.finished = True
        if self.comm is not None:
            self.ctx.comm_destroy(self.comm)
            self.comm = None
        if self.rdzv is not None:
            self.rdzv.close()
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


ROOFLINE_KEYS = ("bound", "served_by", "achieved", "peak", "unit", "frac", "achieved_from_trace", "frac_from_trace", "traced_step_ms", "untraced_step_ms",
                 "traffic", "alg_bytes_per_launch_avg", "avg_launch_ms", "launches_per_step", "kernel_busy_ms", "launches_in_flight",
                 "frac_of_measured_ceiling", "kernel_share_of_step", "device_busy_ms")


def compact_line(full, detail_path):
    """The bench line: numbers, no prose (the driver keeps the TAIL of a line -- the headline's own numbers, the
    roofline, the CPU baseline and the API loops come before the per-configuration legs).  Every key is explained once, in DESIGN.md
    section 6; the verbose record of this run is the --detail file."""
    def pick(d, keys):
        return {k: d[k] for k in keys if k in d}

    def roof(r):
        out = pick(r, ROOFLINE_KEYS)
        out["kernel"] = "k_blur_iter_fast"
        c = r.get("measured_ceiling") or {}
        out["ceiling_GBs"] = {"infinity_cache": c.get("infinity_cache_GBs"), "hbm": c.get("hbm_GBs")}
        return out

    line = pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = full["config"]
    line["config"] = {"workload": f"{cfg['workload'].split(',')[0]}, batch {cfg['global_batch'] // full['n_gpus']} per GPU, Farneback + FoE + phi/threshold + box",
                      "global_batch": cfg["global_batch"], "parallelism": cfg["parallelism"], "comm_ranks": cfg["comm_ranks"],
                      "record_exchange": cfg["record_exchange"], "torch_in_process": cfg["torch_in_process"],
                      "host_cores_per_rank": cfg["host_cores_per_rank"], "runtime": cfg["runtime"]}
    line.update(pick(full, ("hip_event_ms_per_step", "device_busy_ms", "pipeline_alg_bytes_per_pair", "pipeline_frac_of_8TBs", "host_enqueue_ms_per_step",
                            "host_enqueue_share_of_step", "value_incl_h2d", "value_incl_h2d_pipelined", "source_hash", "gathered_rank_blocks_distinct", "rehearsal")))
    if "roofline" in full:
        line["roofline"] = roof(full["roofline"])
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = pick(cb, ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["sample"] = cb["sample"].split(",")[0]
        if "all_cores" in cb:
            line["cpu_baseline"]["all_cores"] = pick(cb["all_cores"], ("value", "cores", "host_cores", "pairs", "seconds"))
    if "api_loop" in full:
        al = full["api_loop"]
        line["api_loop"] = {k: ({kk: vv for kk, vv in v.items() if kk != "flow_seam"} if isinstance(v, dict) else v) for k, v in al.items() if k != "workload"}
    if "video_sequence" in full:
        line["video_sequence"] = pick(full["video_sequence"], ("value", "ms_per_step", "ms_per_step_as_two_batches"))
    if "verification" in full:
        v = full["verification"]
        line["verified_pairs"] = full["verified_pairs"]
        line["verification"] = {"failed_pairs": v["failed_pairs"], "all_pairs_equal_plain_schedule": v.get("all_pairs_equal_plain_schedule"),
                                "flow_epe_px": pick(v["flow_epe_px"], ("mean", "p99.9", "max", "against"))}
    if "configs" in full:
        line["configs"] = {}
        for name, c in full["configs"].items():
            o = pick(c, ("ms_per_call_hip_events", "ms_per_call_wall", "ms_per_pair", "pairs_per_s", "frac", "median_ms", "p99_ms", "max_ms", "latency_calls",
                         "host_enqueue_ms_per_call", "verified_pairs", "failed_pairs", "all_pairs_equal_plain_schedule"))
            if "flow_epe_px" in c:
                o["flow_epe_px"] = pick(c["flow_epe_px"], ("mean", "p99.9", "max"))
            if "roofline" in c:
                o["roofline"] = roof(c["roofline"])
            if "lanes" in c:
                o["lanes"] = {k: v for k, v in c["lanes"].items() if k != "is"}
            line["configs"][name] = o
    if detail_path:
        line["detail"] = detail_path
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (200 x 28 ms: the GPU phase lasts > 5 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--levels", type=int, default=1, help="Farneback levels (BASELINE config 5 uses 5 at 3840x2160)")
    ap.add_argument("--group", type=int, default=0, help="pairs per launch (0 = library default)")
    ap.add_argument("--group-fine", type=int, default=-1, help="pairs per launch for the finest layer's sweeps (-1 = library default)")
    ap.add_argument("--cpu-pairs", type=int, default=16, help="pairs in the one-core CPU baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="mav_set_option before the run (A/B experiments); repeatable")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on ONE GPU (tests): every rank on device 0, records exchanged on the host -- through the rendezvous "
                         "store, or torch's all-gather over gloo with --rendezvous torch (RCCL refuses two ranks on one device); the figures it "
                         "prints are not a scaling measurement")
    ap.add_argument("--rendezvous", choices=("socket", "torch"), default="socket",
                    help="how the ranks of an N > 1 run find each other: socket (default) = mavflow.rendezvous, Python stdlib only, RCCL and the "
                         "HIP runtime from /opt/rocm as libmavflow was built; torch = torch.distributed (its bundled runtime + RCCL).  The socket "
                         "path falls back to the torch path by itself when it does not come up on every rank")
    ap.add_argument("--simulate-socket-failure", action="store_true", help="(tests) the last rank reports that its communicator did not come up: every rank "
                    "must agree to fall back to the torch path")
    ap.add_argument("--no-api-loop", action="store_true", help="skip the reference-shaped loops leg (Processor.run_detection_batched / run_detection on host frames)")
    ap.add_argument("--detail", default=None, help="where the verbose record of the run goes (default: gpurun_out/bench_detail.json when that directory exists)")
    ap.add_argument("--no-configs", action="store_true", help="skip the extra BASELINE configuration legs (C2: 1280x720 batch 1; C5 share: 3840x2160, 5 levels, batch 16)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by a launcher: start the ranks (as children, before anything here touches the GPU) and relay the exit code.
        # Default: bench.py's own launcher -- this process hosts the rendezvous store and spawns one rank per GPU, no torch anywhere;
        # --rendezvous torch: python -m torch.distributed.run, as the driver starts it.
        if args.rendezvous == "torch":
            import socket
            import subprocess
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.call(cmd))
        from mavflow import rendezvous
        sys.exit(rendezvous.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("MAVFLOW_BENCH_DIST") == "1"
    host_cores = pin_rank_to_its_cores(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    ex = RecordExchange(args, rank, world, local_rank) if distributed else None
    if ex is not None and ex.mode == "torch":
        ex.init_torch()                                # torch BEFORE libmavflow: the library then binds to the runtime torch loaded

    import numpy as np
    from mavflow import _lib, synth

    W, H, B = args.width, args.height, args.batch
    ctx = _lib.Context(W, H, B, _lib.fb_defaults(levels=args.levels), device=local_rank)
    if args.group:
        ctx.set_option("group", args.group)
    if args.group_fine >= 0:
        ctx.set_option("group_fine", args.group_fine)
    for kv in args.opt:
        name, val = kv.split("=")
        ctx.set_option(name, int(val))
    layers = [ctx.layer_dims(k)[:2] for k in range(ctx.num_layers())]
    schedule = ctx.schedule_info(B)                    # every option in effect + the per-layer plan of a B-pair call

    prev, nxt = synth.make_batch(W, H, B, distinct=4)
    if rank:                                           # different content per rank
        prev = np.roll(prev, 31 * rank, axis=2); nxt = np.roll(nxt, 31 * rank, axis=2)
    samples = np.stack([synth.foe_samples(W, H, rank * B + b) for b in range(B)])
    d_prev = ctx.alloc(prev.nbytes).upload(prev)
    d_next = ctx.alloc(nxt.nbytes).upload(nxt)
    d_smp = ctx.alloc(samples.nbytes).upload(samples)
    rec = _lib.RESULT_DTYPE.itemsize
    d_res = ctx.alloc(B * rec)                         # this rank's records
    d_mf = ctx.alloc(B * W * H)
    d_md = ctx.alloc(B * W * H)

    # the record exchange: RCCL all-gather on the CONTEXT's stream (mav_allgather_results), no host synchronisation inside a step.
    if ex is not None:
        if not ex.connect(ctx, d_res, B * rec):
            # the torch-free path did not come up on every rank: run the whole measurement again in a CHILD process on the
            # torch.distributed path (a fresh process: torch's runtime first), relay its output and exit code
            for d in (d_prev, d_next, d_smp, d_res, d_mf, d_md):
                d.free()
            ctx.close()
            import subprocess
            argv = [a for a in sys.argv[1:] if a not in ("--rendezvous", "socket", "--simulate-socket-failure")] + ["--rendezvous", "torch"]
            code = subprocess.call([sys.executable, os.path.abspath(__file__)] + argv)
            sys.stdout.flush()
            os._exit(code)                             # (a rank stuck inside RCCL's bootstrap may have left a thread behind)
    exchange = ex.describe() if ex is not None else "none (1 GPU)"

    def run_batch():
        # flow stays in the library's HBM workspace (flow_ptr=None); both threshold masks are written out (1 B/px each),
        # the per-pair box + FoE records are the result
        ctx.process_batch_dev(d_prev.ptr, d_next.ptr, d_smp.ptr, B, d_res.ptr, mf_ptr=d_mf.ptr, md_ptr=d_md.ptr)

    def step():
        run_batch()
        if ex is not None:
            ex.exchange()                              # stream-ordered behind the batch, ahead of the next one (library communicator)

    def barrier():
        ctx.sync()
        if ex is not None:
            ex.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    enq_loop = time.perf_counter() - t0               # every step is enqueued (includes any back-pressure of a full queue); nothing synchronised yet
    ev_ms = ctx.timer_stop()
    barrier()
    elapsed = time.perf_counter() - t0
    if ex is not None:
        elapsed = ex.max_over_ranks(elapsed)

    # ---- what did the timed loop compute?  (outside the timed region) ----
    verification = None
    res_last = d_res.download(_lib.RESULT_DTYPE, (B,))
    gathered_distinct = comm_ranks = None
    if ex is not None:                                 # the gathered block of this rank must be its own records
        allrec = ex.gathered().view(_lib.RESULT_DTYPE)
        assert allrec[rank * B:(rank + 1) * B].tobytes() == res_last.tobytes(), "all-gathered records differ from the local ones"
        if world > 1:                                  # every rank runs different content (frames rolled by 31 px per rank, other samples)
            gathered_distinct = len({allrec[r * B:(r + 1) * B].tobytes() for r in range(world)})
        comm_ranks = ex.comm_ranks()
        # the ranks part here: the communicator goes, ranks other than 0 leave; rank 0's post-loop legs (verification, H2D, video,
        # roofline) hold nobody
        ex.finish()
        if rank != 0:
            ctx.close()
            return
    if rank == 0 and not args.no_verify:
        verification = verify_last_step(ctx, prev, nxt, samples, res_last, d_mf, d_md, sorted({0, B - 1}), args.levels)
        # ... and EVERY pair of the timed step against the same batch re-run in the plain schedule
        ok, n_same, differing = equal_plain_schedule(ctx, run_batch, d_res, d_mf, d_md, B, W, H)
        verification["all_pairs_equal_plain_schedule"] = ok
        verification["pairs_with_identical_flow_in_plain_schedule"] = n_same
        if not ok:
            verification["failed_pairs"] = sorted(set(verification["failed_pairs"]) | set(differing))

    # ---- PCIe-inclusive rates (never `value`).  (1) naive: the two u8 frame stacks uploaded synchronously from pageable memory
    #      inside the loop; (2) pipelined: pinned memory, uploads on the context's copy stream into a second buffer set while
    #      the previous batch computes (how a deployment would feed the GPU) ----
    h2d_ms = h2d_pipe_ms = None
    if rank == 0 and not args.no_profile:
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(2):
            d_prev.upload(prev); d_next.upload(nxt)
            run_batch()
        ctx.sync()
        h2d_ms = 1e3 * (time.perf_counter() - t1) / 2
        hp, hn = ctx.pinned_like(prev), ctx.pinned_like(nxt)
        sets = [(d_prev, d_next), (ctx.alloc(prev.nbytes), ctx.alloc(nxt.nbytes))]
        ctx.upload_async(sets[0][0], hp); ctx.upload_async(sets[0][1], hn); ctx.upload_fence()
        ctx.sync()
        n_pipe = 4
        t1 = time.perf_counter()
        for k in range(n_pipe):
            cur, nx = sets[k & 1], sets[(k + 1) & 1]
            ctx.upload_async(nx[0], hp); ctx.upload_async(nx[1], hn)              # next batch crosses PCIe now (behind batch k-1) ...
            ctx.process_batch_dev(cur[0].ptr, cur[1].ptr, d_smp.ptr, B, d_res.ptr, mf_ptr=d_mf.ptr, md_ptr=d_md.ptr)   # ... while this one computes
            ctx.upload_fence()
        ctx.sync()
        h2d_pipe_ms = 1e3 * (time.perf_counter() - t1) / n_pipe

    # ---- the same path on a VIDEO (never `value`): B + 1 consecutive frames, pair i = (frame i, frame i + 1), handed over as two
    #      views of one buffer.  The library recognises the layout and blurs / expands every frame once per group instead of twice
    #      (flow bit-identical to the two-batch form: tests/test_gpu_flow.py); the headline metric stays on independent pairs ----
    video = None
    if rank == 0 and not args.no_profile:
        seq = synth.make_sequence(W, H, B + 1)
        d_seq = ctx.alloc(seq.nbytes).upload(seq)
        n_vid = max(5, min(30, args.steps))
        for _ in range(2):
            ctx.process_batch_dev(d_seq.ptr, d_seq.ptr + W * H, d_smp.ptr, B, d_res.ptr, mf_ptr=d_mf.ptr, md_ptr=d_md.ptr)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(n_vid):
            ctx.process_batch_dev(d_seq.ptr, d_seq.ptr + W * H, d_smp.ptr, B, d_res.ptr, mf_ptr=d_mf.ptr, md_ptr=d_md.ptr)
        ctx.sync()
        vid_ms = 1e3 * (time.perf_counter() - t1) / n_vid
        # the same frames as two separate batches (nothing to recognise): what the sharing itself is worth on this content
        d_p2, d_n2 = ctx.alloc(B * W * H).upload(seq[:-1]), ctx.alloc(B * W * H).upload(seq[1:])
        ctx.process_batch_dev(d_p2.ptr, d_n2.ptr, d_smp.ptr, B, d_res.ptr, mf_ptr=d_mf.ptr, md_ptr=d_md.ptr)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(n_vid):
            ctx.process_batch_dev(d_p2.ptr, d_n2.ptr, d_smp.ptr, B, d_res.ptr, mf_ptr=d_mf.ptr, md_ptr=d_md.ptr)
        ctx.sync()
        two_ms = 1e3 * (time.perf_counter() - t1) / n_vid
        video = {"value": round(B / (vid_ms * 1e-3), 2), "unit": "frame-pairs/s", "ms_per_step": round(vid_ms, 3), "steps": n_vid,
                 "ms_per_step_as_two_batches": round(two_ms, 3),
                 "workload": f"{B + 1} consecutive synthetic frames = {B} pairs sharing their inner frames (next = prev + one frame)"}
        for d in (d_seq, d_p2, d_n2):
            d.free()

    # ---- roofline of the dominant kernel (separate passes, HIP events on the streams the launches go to) and the host side of a step ----
    roofline = None
    ceil = None
    enqueue_ms = None
    if rank == 0 and not args.no_profile:
        enqueue_ms = host_enqueue_ms(ctx, step)
        ceil = measured_ceilings(ctx)
        roofline = sweep_roofline(ctx, run_batch, B, layers, W, H, args.levels, schedule, 1e3 * elapsed / args.steps, ceil)

    # ---- the other BASELINE configurations one GPU can hold (never `value`; outside the headline's timed region) ----
    configs = None
    extra_legs = rank == 0 and world == 1 and (W, H, args.levels) == (1920, 1080, 1)
    if extra_legs and not (args.no_configs and args.no_api_loop):
        # the headline's buffers and its whole context (2.9 GB of workspace, 8.5 GB of flow) are released first: the legs then get the
        # memory a stand-alone run of their configuration would get
        for d in (d_prev, d_next, d_mf, d_md, d_smp, d_res):
            d.free()
        ctx.close()
    if extra_legs and not args.no_configs:
        configs = {"C2": run_config_leg("C2", 1280, 720, 1, 1, 300, [0], verify=not args.no_verify, ceil=ceil),
                   "C5_share": run_config_leg("C5_share", 3840, 2160, 16, 5, 10, [0, 15], verify=not args.no_verify, ceil=ceil)}
        configs["C5_share"]["note"] = "per-GPU share of BASELINE config 5 (batch 128 across 8 GPUs); its CPU baseline (73 s) is not repeated here"

    # ---- the reference-shaped loops (never `value`): Processor.run_detection_batched / run_detection on host numpy frames ----
    api_loop = None
    if extra_legs and not args.no_api_loop:
        api_loop = api_loop_leg(W, H, batch=B)

    failed = False
    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / elapsed
        balg = b_alg_per_pair(layers, W, H, ctx.fb.iterations)
        # `full`: everything this run measured, with the prose that says how (written to --detail, default gpurun_out/bench_detail.json);
        # the ONE line on stdout carries the numbers only -- what each key means is in DESIGN.md section 6 and profiles/README.md
        full = {"metric": "frame-pairs/sec at 1920x1080" if (W, H) == (1920, 1080) else f"frame-pairs/sec at {W}x{H}",
                "value": round(value, 2), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{W}x{H}, batch={B} frame pairs per GPU, Farneback(0.4,{ctx.fb.levels},12,10,8,1.2,0) "
                                       f"+ FoE(1000 pairs) + phi/threshold + box, {len(layers)} pyramid layers",
                           "global_batch": world * B, "parallelism": f"frame-parallel x{world}" + (", all-gather of 32-B records" if world > 1 else ""),
                           "record_exchange": exchange, "comm_ranks": comm_ranks, "torch_in_process": "torch" in sys.modules,
                           "host_cores_per_rank": host_cores, "schedule": schedule, "runtime": _lib.runtime_info()},
                "hip_event_ms_per_step": round(ev_ms / args.steps, 3),
                "pipeline_alg_bytes_per_pair": balg,
                "pipeline_alg_GBs": round(value / world * balg / 1e9, 1),
                "pipeline_frac_of_8TBs": round(value / world * balg / (HBM_PEAK_GBS * 1e9), 4),
                "source_hash": source_hash(schedule), "kernel_source_hash": source_hash()}
        full["host_enqueue_ms_per_step_in_loop"] = round(1e3 * enq_loop / args.steps, 3)
        if enqueue_ms is not None:
            full["host_enqueue_ms_per_step"] = round(enqueue_ms, 3)
            full["host_enqueue_share_of_step"] = round(enqueue_ms / (1e3 * elapsed / args.steps), 3)
        if gathered_distinct is not None:
            full["gathered_rank_blocks_distinct"] = gathered_distinct
        if args.rehearse_on_one_gpu:
            full["rehearsal"] = f"{world} ranks share ONE GPU, records exchanged on the host: a functional run of the N > 1 path, not a scaling measurement"
        if h2d_ms:
            full["value_incl_h2d"] = round(B / (h2d_ms * 1e-3), 2)
            full["value_incl_h2d_pipelined"] = round(B / (h2d_pipe_ms * 1e-3), 2)
        if roofline:
            full["device_busy_ms"] = roofline["device_busy_ms"]
            full["roofline"] = roofline
        if world == 1 and args.cpu_pairs > 0:
            full["cpu_baseline"] = cpu_baseline(prev, nxt, samples, min(args.cpu_pairs, B), args.levels)
        if api_loop:
            full["api_loop"] = api_loop
            failed = failed or not api_loop["batched_and_unbatched_results_identical"]
        if video:
            full["video_sequence"] = video
        if verification:
            full.update({"verified_pairs": verification["verified_pairs"], "verification": {k: v for k, v in verification.items() if k != "verified_pairs"}})
            failed = failed or bool(verification["failed_pairs"])
        if configs:
            full["configs"] = configs
            failed = failed or any(c.get("failed_pairs") for c in configs.values())
        detail = args.detail
        if detail is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
            detail = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        if detail:
            try:
                with open(detail, "w") as f:
                    json.dump(full, f, indent=1)
            except OSError:
                detail = None
        print(json.dumps(compact_line(full, os.path.relpath(detail, ROOT) if detail else None)), flush=True)
    ctx.close()
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
