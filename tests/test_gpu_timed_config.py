"""What bench.py times is what these tests verify: mav_process_batch_dev with device pointers, flow = NULL (the flow stays in the
context's workspace), phi = NULL (the single-precision screen of the phi kernel is ON), library-default scheduling (8 pairs per
launch, the finest layer's sweeps one pair per launch) -- at the BASELINE configurations' own sizes and batch sizes
(/root/reference/src/processor.py:305-341 is the loop body being replaced):
    C3  1920x1080, batch 64     records and both masks of pairs {0, 7, 8, 37, 63} bit for bit against the oracle chain on the
                                flow the timed call itself left in the workspace; that flow against the C oracle (EPE gate)
    C2  1280x720,  batch 1      the whole chain
    C5  3840x2160, levels = 5, batch 16 (one GPU's share)   oracle on one pair; box == extents of the mask and determinism on all 16
"""
import numpy as np
import pytest

from oracle import foe_oracle as fo
from oracle.tolerances import check_flow
from mavflow import synth

pytestmark = pytest.mark.gpu


class TimedRun:
    """bench.py's run_batch(), plus downloads of what it leaves behind."""

    def __init__(self, ctx, prev, nxt, smp):
        from mavflow import _lib
        self.ctx, self.B = ctx, prev.shape[0]
        self.H, self.W = prev.shape[1:]
        self.d_prev = ctx.alloc(prev.nbytes).upload(prev)
        self.d_next = ctx.alloc(nxt.nbytes).upload(nxt)
        self.d_smp = ctx.alloc(smp.nbytes).upload(smp)
        self.d_res = ctx.alloc(self.B * _lib.RESULT_DTYPE.itemsize)
        self.d_mf = ctx.alloc(self.B * self.W * self.H)
        self.d_md = ctx.alloc(self.B * self.W * self.H)
        self.dtype = _lib.RESULT_DTYPE

    def run(self):
        self.ctx.process_batch_dev(self.d_prev.ptr, self.d_next.ptr, self.d_smp.ptr, self.B, self.d_res.ptr,
                                   mf_ptr=self.d_mf.ptr, md_ptr=self.d_md.ptr)
        self.ctx.sync()
        res = self.d_res.download(self.dtype, (self.B,))
        mf = self.d_mf.download(np.uint8, (self.B, self.H, self.W)).view(np.bool_)
        md = self.d_md.download(np.uint8, (self.B, self.H, self.W)).view(np.bool_)
        return res, mf, md

    def free(self):
        for b in (self.d_prev, self.d_next, self.d_smp, self.d_res, self.d_mf, self.d_md):
            b.free()


def check_pair_against_oracle(ctx, b, smp_b, res, mf, md):
    """records + both masks of pair b, bit for bit, against the numpy chain on the flow the timed call left in the workspace"""
    flow = ctx.last_flow(b)
    chain = fo.run_chain(flow, smp_b)
    assert tuple(res[b]["foe"]) == tuple(chain["foe"]), (b, tuple(res[b]["foe"]), chain["foe"])
    assert np.array_equal(mf[b], chain["fixed"]), (b, int((mf[b] != chain["fixed"]).sum()))
    assert np.array_equal(md[b], chain["total"]), (b, int((md[b] != chain["total"]).sum()))
    assert tuple(res[b]["box"]) == tuple(chain["box"]), (b, tuple(res[b]["box"]), chain["box"])
    return flow


def test_c3_1080p_batch64_as_timed(mav, fb_oracle):
    from mavflow import _lib
    W, H, B = 1920, 1080, 64
    prev, nxt = synth.make_batch(W, H, B, distinct=4)                 # bench.py's batch
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
    with _lib.Context(W, H, B) as ctx:
        t = TimedRun(ctx, prev, nxt, smp)
        t.run()                                                        # warm-up, as the bench does
        res, mf, md = t.run()
        for b in (0, 7, 8, 37, 63):
            flow = check_pair_against_oracle(ctx, b, smp[b], res, mf, md)
            if b in (0, 37):
                check_flow(flow, fb_oracle.calc(prev[b], nxt[b]), b)
        for b in range(B):                                             # box == extents of the fixed mask, every pair
            assert tuple(res[b]["box"]) == fo.simple_bounding_box(mf[b]), b
        # shifted copies of a base pair (make_batch) see the FoE shifted by the same (13 s, 7 s): a second, oracle-free check
        # that slot b really holds pair b's result and not a neighbour's
        assert abs(res[0]["foe"][0] - 0.55 * W) < 30 and abs(res[0]["foe"][1] - 0.45 * H) < 30
        res2, mf2, md2 = t.run()                                       # the timed loop repeats the call: must be deterministic
        assert res2.tobytes() == res.tobytes() and np.array_equal(mf2, mf) and np.array_equal(md2, md)
        t.free()


def test_c2_720p_batch1_full_chain(mav, fb_oracle):
    from mavflow import _lib
    W, H = 1280, 720
    f0, f1, _ = synth.make_pair(W, H, 1)
    prev, nxt = f0[None], f1[None]
    smp = synth.foe_samples(W, H, 5)[None]
    with _lib.Context(W, H, 1) as ctx:
        t = TimedRun(ctx, prev, nxt, smp)
        res, mf, md = t.run()
        flow = check_pair_against_oracle(ctx, 0, smp[0], res, mf, md)
        check_flow(flow, fb_oracle.calc(f0, f1), "C2")
        assert mf[0].any() and md[0].any()
        assert abs(res[0]["foe"][0] - 0.55 * W) < 30 and abs(res[0]["foe"][1] - 0.45 * H) < 30
        # the host-pointer entry point (what the reference-shaped loop calls) gives the same answer
        out = ctx.process_batch(prev, nxt, smp)
        assert out["results"].tobytes() == res.tobytes() and np.array_equal(out["flow"][0], flow)
        assert np.array_equal(out["mask_fixed"], mf) and np.array_equal(out["mask_dyn"], md)
        t.free()


def test_c5_4k_five_layers_batch16_full_chain(mav, fb_oracle):
    from mavflow import _lib
    from oracle import fb_oracle as fbo
    W, H, B = 3840, 2160, 16
    base = [synth.make_pair(W, H, 7 + i, k=0.004)[:2] for i in range(2)]
    prev = np.stack([np.roll(base[b % 2][0], (5 * (b // 2), 11 * (b // 2)), axis=(0, 1)) for b in range(B)])
    nxt = np.stack([np.roll(base[b % 2][1], (5 * (b // 2), 11 * (b // 2)), axis=(0, 1)) for b in range(B)])
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
    with _lib.Context(W, H, B, _lib.fb_defaults(levels=5)) as ctx:
        assert ctx.num_layers() == 5
        t = TimedRun(ctx, prev, nxt, smp)
        res, mf, md = t.run()
        flow = check_pair_against_oracle(ctx, 3, smp[3], res, mf, md)
        check_flow(flow, fb_oracle.calc(prev[3], nxt[3], fbo.default_params(levels=5)), "C5 share")
        check_pair_against_oracle(ctx, 12, smp[12], res, mf, md)       # a pair from the middle of the batch
        for b in range(B):
            assert tuple(res[b]["box"]) == fo.simple_bounding_box(mf[b]), b
        res2, mf2, md2 = t.run()
        assert res2.tobytes() == res.tobytes() and np.array_equal(mf2, mf) and np.array_equal(md2, md)
        t.free()
