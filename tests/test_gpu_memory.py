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
        # a stage hook that needs the two-pass scratch takes a staging block of its own: no Farneback workspace, and "deep_frac"
        # (settable until the first call that computes flow) stays open
    with _lib.Context(W, H, 1) as ctx:
        a = ctx.stage_blur_resize(prev[0], 1, two_pass=True)
        assert ctx.mem_info()["workspace_bytes"] == 0
        assert np.array_equal(a, ctx.stage_blur_resize(prev[0], 1))
        ctx.set_option("deep_frac", 32)


def test_deep_layer_workspace_arrives_with_the_first_call_of_more_than_one_group(mav):
    """The deep layers' work set is reachable only by calls of more than one group: a context whose calls stay within one group never
    holds it, and mav_mem_info's workspace figure grows by exactly that set when the first such call comes."""
    from mavflow import _lib
    W, H, B = 320, 240, 6
    prev, nxt = synth.make_batch(W, H, B, distinct=2)
    with _lib.Context(W, H, B) as ctx:
        ctx.set_option("group", 2)
        one = ctx.farneback(prev[:2], nxt[:2]).copy()            # one group: no deep set
        ws1 = ctx.mem_info()["workspace_bytes"]
        all6 = ctx.farneback(prev, nxt)                           # three groups: the deep layers run once for the call
        ws2 = ctx.mem_info()["workspace_bytes"]
        assert ws2 > ws1 > 0
        assert np.array_equal(all6[:2], one)
        ctx.farneback(prev, nxt)
        assert ctx.mem_info()["workspace_bytes"] == ws2


def test_helper_context_cache_is_bounded_and_a_held_context_survives_eviction(mav):
    """The helpers' context cache keeps the most recently used frame sizes and only DROPS what falls out -- it never closes a context,
    because a caller may still hold it: pyramid() across its yields, a loop across its frames."""
    from mavflow import im_helpers
    im_helpers._ctx_cache.clear()
    cap = im_helpers._CTX_CACHE_SIZES
    sizes = [(64 + 16 * k, 48 + 12 * k) for k in range(cap + 2)]
    seen = []
    for (W, H) in sizes:
        img = np.zeros((H, W), np.uint8)
        img[5:9, 7:20] = 200
        r = im_helpers.get_simple_bounding_box(img)
        assert r.get_topleft() == (7, 5)
        seen.append(im_helpers._ctx_cache[(W, H)])
    assert list(im_helpers._ctx_cache) == sizes[2:]               # `cap` sizes kept, the two oldest dropped ...
    assert seen[0].h is not None                                  # ... but NOT closed: the handle we hold still works
    img = np.zeros((sizes[0][1], sizes[0][0]), np.uint8)
    img[3, 4] = 9
    assert tuple(seen[0].bbox(img)[0]) == (4, 3, 4, 3)
    im_helpers.get_simple_bounding_box(np.zeros((sizes[3][1], sizes[3][0]), np.uint8))
    assert list(im_helpers._ctx_cache)[-1] == sizes[3]            # a hit moves to the back
    c = im_helpers._ctx(*sizes[4], batch=4)                       # a larger batch replaces the cached context, the old one stays usable
    assert c.max_batch == 4 and seen[4].h is not None and c is not seen[4]
    im_helpers._ctx_cache.clear()


def test_pyramid_generator_survives_helper_calls_on_every_level_size(mav):
    """pyramid() holds its context across yields; calling a helper on each level's own size (more sizes than the cache keeps would
    evict that context) must not break the next level."""
    from mavflow import im_helpers
    im_helpers._ctx_cache.clear()
    keep = im_helpers._CTX_CACHE_SIZES
    im_helpers._CTX_CACHE_SIZES = 2
    try:
        img = (np.add.outer(np.arange(540), np.arange(960)) % 251).astype(np.uint8)
        levels = []
        for lv in im_helpers.pyramid(img, scale=1.5):
            box = im_helpers.get_simple_bounding_box(lv)          # a context of this level's size: evicts the generator's
            assert box.get_topleft()[0] >= 0
            levels.append(lv.shape)
        assert len(levels) >= 6 and levels[0] == (540, 960)
    finally:
        im_helpers._CTX_CACHE_SIZES = keep
        im_helpers._ctx_cache.clear()


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
