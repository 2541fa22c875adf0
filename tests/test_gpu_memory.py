"""Device-memory behaviour of a context (include/mavflow.h: mav_mem_info): the Farneback workspace belongs to the calls that compute
flow, the reference's stateless helpers (src/im_helpers.py:55-84, 244-252) must not pay for it; the Python helpers keep a bounded
number of contexts."""
import numpy as np
import pytest

from mavflow import synth

pytestmark = pytest.mark.gpu
MB = 1 << 20


def test_a_4k_context_used_for_bbox_only_takes_a_few_megabytes(mav):
    from mavflow import _lib
    W, H = 3840, 2160
    with _lib.Context(64, 48, 1) as warm:                        # the runtime's own first-use allocations (code objects, pools)
        warm.bbox(np.zeros((48, 64), np.uint8))
        free0 = warm.mem_info()["dev_free"]
        img = np.zeros((H, W), np.uint8)
        img[700:720, 1000:1100] = 255
        with _lib.Context(W, H, 1) as ctx:
            assert tuple(ctx.bbox(img)[0]) == (1000, 700, 1099, 719)
            info = ctx.mem_info()
            taken = free0 - info["dev_free"]
            print(f"3840x2160 context after bbox: {taken / MB:.1f} MB of device memory taken, ctx_bytes {info['ctx_bytes'] / MB:.1f} MB, "
                  f"workspace {info['workspace_bytes']}")
            assert info["workspace_bytes"] == 0
            assert taken < 64 * MB and info["ctx_bytes"] < 64 * MB
            # validation counts and the detection tail do not bring it either
            ctx.tpr_fpr_counts(img, (img > 0).astype(np.uint8))
            assert ctx.mem_info()["workspace_bytes"] == 0


def test_the_workspace_comes_with_the_first_flow_call_and_stays(mav):
    from mavflow import _lib
    W, H, B = 640, 480, 2
    prev, nxt = synth.make_batch(W, H, B, distinct=B)
    with _lib.Context(W, H, B) as ctx:
        assert ctx.mem_info()["workspace_bytes"] == 0
        sched = ctx.schedule_info(B)                              # planning needs no workspace
        assert sched["pairs_per_group"] == B
        ctx.set_option("group", 1)                                # ... nor does changing the plan
        assert ctx.mem_info()["workspace_bytes"] == 0
        ctx.set_option("group", 2)
        f1 = ctx.farneback(prev, nxt).copy()
        ws = ctx.mem_info()["workspace_bytes"]
        assert ws > 0
        f2 = ctx.farneback(prev, nxt)
        assert ctx.mem_info()["workspace_bytes"] == ws and np.array_equal(f1, f2)
        # a stage hook that needs the two-pass scratch allocates on a fresh context too
    with _lib.Context(W, H, 1) as ctx:
        a = ctx.stage_blur_resize(prev[0], 1, two_pass=True)
        assert ctx.mem_info()["workspace_bytes"] > 0
        assert np.array_equal(a, ctx.stage_blur_resize(prev[0], 1))


def test_helper_context_cache_is_bounded_and_closes_what_it_drops(mav):
    from mavflow import im_helpers
    im_helpers._ctx_cache.clear()
    sizes = [(64, 48), (80, 60), (96, 72)]
    seen = []
    for (W, H) in sizes:
        img = np.zeros((H, W), np.uint8)
        img[5:9, 7:20] = 200
        r = im_helpers.get_simple_bounding_box(img)
        assert r.get_topleft() == (7, 5)
        seen.append(im_helpers._ctx_cache[(W, H)])
    assert list(im_helpers._ctx_cache) == sizes[1:]               # two sizes kept, the oldest dropped ...
    assert seen[0].h is None                                      # ... and closed
    im_helpers.get_simple_bounding_box(np.zeros((60, 80), np.uint8))
    assert list(im_helpers._ctx_cache) == [sizes[2], sizes[1]]    # a hit moves to the back
    c = im_helpers._ctx(96, 72, batch=4)                          # a larger batch replaces (and closes) the cached context
    assert c.max_batch == 4 and seen[2].h is None


def test_two_contexts_in_two_threads_are_independent(mav):
    """include/mavflow.h: a mav_ctx is single-threaded, distinct contexts are independent.  Two host threads, each with a context of its
    own (different frame sizes, hence different streams, workspaces, tables and option sets), run the fused path concurrently; every
    result must equal what the same context gives alone."""
    import threading
    from mavflow import _lib
    jobs = []
    for (W, H, B, levels) in ((640, 480, 3, 1), (1000, 562, 2, 3)):
        prev, nxt = synth.make_batch(W, H, B, distinct=B)
        smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
        with _lib.Context(W, H, B, _lib.fb_defaults(levels=levels)) as c:
            ref = c.process_batch(prev, nxt, smp)
        jobs.append((W, H, B, levels, prev, nxt, smp, ref))
    errors = []

    def work(job):
        W, H, B, levels, prev, nxt, smp, ref = job
        try:
            with _lib.Context(W, H, B, _lib.fb_defaults(levels=levels)) as c:
                c.set_option("group", 1 + (W % 2))
                for rep in range(6):
                    out = c.process_batch(prev, nxt, smp)
                    assert np.array_equal(out["flow"], ref["flow"]), (W, rep)
                    assert out["results"].tobytes() == ref["results"].tobytes(), (W, rep)
                    assert np.array_equal(out["mask_fixed"], ref["mask_fixed"]) and np.array_equal(out["mask_dyn"], ref["mask_dyn"])
        except Exception as e:                         # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
