"""Shape fuzz: random frame sizes / Farneback parameters through the C-ABI against the oracle (run on the GPU box).
usage: python tools/fuzz_shapes.py [n_cases] [seed]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np


def fuzz_cases(n_cases, seed):
    """The fuzz's cases as dicts (W, H, B, fb, po, prev, nxt, smp, group), drawn from ONE generator in a fixed order so that
    (seed, case) names a case for good (tools/worst_pixel.py, tests/test_gpu_flow.py's regression case replay them)."""
    from mavflow import _lib, synth
    from oracle import fb_oracle
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        if case % 3 == 0:
            W, H = int(rng.integers(8, 64)) * 4, int(rng.integers(33, 200))          # fast-path widths (multiples of 4)
        elif case % 3 == 1:
            W, H = int(rng.integers(33, 300)), int(rng.integers(33, 200))            # arbitrary widths (relaxed-alignment kernels)
        else:
            W, H = int(rng.choice([64, 128, 192, 256, 320, 704])), int(rng.choice([48, 64, 96, 160, 208]))
        fb = _lib.fb_defaults()
        po = fb_oracle.default_params()
        if case % 5 == 4:                                                              # non-default parameters
            fb.pyr_scale = po.pyr_scale = float(rng.choice([0.5, 0.6, 0.4]))
            fb.levels = po.levels = int(rng.integers(0, 4))
            fb.winsize = po.winsize = int(rng.choice([5, 9, 12, 13, 15, 21]))
            fb.iterations = po.iterations = int(rng.integers(1, 5))
            fb.poly_n = po.poly_n = int(rng.choice([5, 7, 8]))
            fb.poly_sigma = po.poly_sigma = float(rng.choice([1.1, 1.2, 1.5]))
        B = int(rng.integers(1, 4))
        if case % 11 == 10:                                                            # a many-layer pyramid on a larger frame, several groups per call
            W, H = int(rng.integers(100, 400)) * 4, int(rng.integers(400, 1000))      # (deep layers once per call, two-pass blur chunks, 64 x 8 tiles)
            fb.pyr_scale = po.pyr_scale = float(rng.choice([0.4, 0.5]))
            fb.levels = po.levels = int(rng.integers(3, 6))
            B = int(rng.integers(3, 7))
        prev = rng.integers(0, 256, (B, H, W)).astype(np.uint8) if case % 7 == 6 else None
        if prev is None:
            pairs = [synth.make_pair(W, H, case * 10 + b, k=0.02, patch=False)[:2] for b in range(B)]
            prev = np.stack([p[0] for p in pairs]); nxt = np.stack([p[1] for p in pairs])
        else:
            nxt = np.roll(prev, (1, 2), axis=(1, 2))
        smp = np.zeros((B, 2000, 2), np.uint32)
        smp[..., 0] = rng.integers(0, H, (B, 2000)); smp[..., 1] = rng.integers(0, W, (B, 2000))
        group = int(rng.integers(1, B + 1)) if B > 1 else 0
        yield dict(case=case, W=W, H=H, B=B, fb=fb, po=po, prev=prev, nxt=nxt, smp=smp, group=group)


def main():
    from mavflow import _lib
    from oracle import fb_oracle, foe_oracle
    from oracle.tolerances import check_flow, unstable_mask, flow_gate, epe
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    orc = fb_oracle.load()
    worst, n_ill, n_excused = 0.0, 0, 0
    for cs in fuzz_cases(n_cases, seed):
        case, W, H, B, fb, po, prev, nxt, smp = (cs[k] for k in ("case", "W", "H", "B", "fb", "po", "prev", "nxt", "smp"))
        t0 = time.time()
        with _lib.Context(W, H, B, fb) as c:
            out = c.process_batch(prev, nxt, smp, want_phi=True)
            # the same frames as ONE run (a frame sequence: every frame expanded once) must give the two-batch flow bit for bit
            run = np.concatenate([prev, nxt[-1:]])
            if B > 1:
                c.set_option("group", cs["group"])
            # ... and the call cut into groups differently (several groups: the deep layers then run once per call) the one-group flow
            assert np.array_equal(c.farneback(prev, nxt), out["flow"]), (case, W, H, "groups")
            two = c.farneback(run[:-1].copy(), run[1:].copy())
            assert np.array_equal(c.farneback_sequence(run), two), (case, W, H, "sequence")
        for b in range(B):
            ref = orc.calc(prev[b], nxt[b], po)
            e = epe(out["flow"][b], ref)
            if flow_gate(e) is not None:                             # outside the strict gate: is it where the oracle itself is unstable?
                ref, _, flips = orc.calc_tracked(prev[b], nxt[b], po)
                twins = orc.twins(prev[b], nxt[b], po)
                e = check_flow(out["flow"][b], ref, (case, W, H, b), twins, fb.winsize // 2, flips)
                n_excused += 1
                n_ill += int(unstable_mask(ref, twins, fb.winsize // 2, flips).sum())
            worst = max(worst, float(e.max()))
            ch = foe_oracle.run_chain(out["flow"][b], smp[b])
            r = out["results"][b]
            assert tuple(r["foe"]) == tuple(ch["foe"]), (case, W, H, tuple(r["foe"]), ch["foe"])
            assert np.array_equal(out["mask_fixed"][b], ch["fixed"]) and np.array_equal(out["mask_dyn"][b], ch["total"]), (case, W, H)
            assert tuple(r["box"]) == tuple(ch["box"]), (case, W, H)
        print(f"case {case:3d}  {W:4d}x{H:<4d} B={B} layers={fb.levels} win={fb.winsize} it={fb.iterations} n={fb.poly_n}  ok ({time.time() - t0:.2f}s)", flush=True)
    print(f"all {n_cases} cases passed; worst single-pixel EPE {worst:.3e} px; {n_excused} frames outside the strict gate, {n_ill} unstable pixels excused (oracle/tolerances.py)")


if __name__ == "__main__":
    main()
