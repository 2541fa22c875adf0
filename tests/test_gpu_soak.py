"""The 8-GPU configurations' GLOBAL sizes on one GPU (BASELINE configs 4 and 5: 1920x1080 x 512 pairs, 3840x2160 / 5 layers x 128
pairs), through mav_process_batch_dev exactly as bench.py calls it (flow = NULL: the 8.5 GB flow workspace is the context's; both
masks out; 32-byte records).  These are the largest contexts include/mavflow.h allows (max_batch * W * H is 1 % under the 2^30-pixel
bound; the flow workspace is 2.12e9 floats): every per-batch offset, the 32 groups of the 1080p call, the deep layers' 64-pair chunks
of the 4K call.  Oracle on pairs {0, middle, last}; box == extents of the fixed mask and determinism on ALL pairs; device memory
printed.  The loop being replaced: /root/reference/src/processor.py:305-341 over a whole sequence."""
import zlib

import numpy as np
import pytest

from oracle import foe_oracle as fo
from oracle.tolerances import check_flow
from mavflow import synth

pytestmark = pytest.mark.gpu
GB = float(1 << 30)


def soak(W, H, B, levels, fb_oracle, tag):
    from mavflow import _lib
    from oracle import fb_oracle as fbo
    assert B * W * H <= 1 << 30
    # four base pairs, every pair of the batch a different shift of one of them (so that slot b cannot be mistaken for a neighbour)
    base = [synth.make_pair(W, H, 11 + i, k=0.01 if W < 3000 else 0.004)[:2] for i in range(4)]
    shift = lambda a, b: np.roll(a, (3 * (b // 4) % H, 7 * (b // 4) % W), axis=(0, 1))
    prev = np.empty((B, H, W), np.uint8)
    nxt = np.empty((B, H, W), np.uint8)
    for b in range(B):
        prev[b], nxt[b] = shift(base[b % 4][0], b), shift(base[b % 4][1], b)
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
    par = fbo.default_params(levels=levels)
    with _lib.Context(W, H, B, _lib.fb_defaults(levels=levels)) as ctx:
        free0 = ctx.mem_info()["dev_free"]
        d_prev, d_next = ctx.alloc(prev.nbytes).upload(prev), ctx.alloc(nxt.nbytes).upload(nxt)
        d_smp = ctx.alloc(smp.nbytes).upload(smp)
        d_res, d_mf, d_md = ctx.alloc(B * _lib.RESULT_DTYPE.itemsize), ctx.alloc(B * W * H), ctx.alloc(B * W * H)

        def run():
            ctx.process_batch_dev(d_prev.ptr, d_next.ptr, d_smp.ptr, B, d_res.ptr, mf_ptr=d_mf.ptr, md_ptr=d_md.ptr)
            ctx.sync()
            return (d_res.download(_lib.RESULT_DTYPE, (B,)), d_mf.download(np.uint8, (B, H, W)).view(np.bool_),
                    d_md.download(np.uint8, (B, H, W)).view(np.bool_))
        res, mf, md = run()
        info = ctx.mem_info()
        sched = ctx.schedule_info(B)
        print(f"{tag}: {B} pairs in groups of {sched['pairs_per_group']}, deep layers from {sched['deep_layers_from']} x {sched['deep_pairs']} pairs; "
              f"device memory taken {(free0 - info['dev_free']) / GB:.2f} GB, context {info['ctx_bytes'] / GB:.2f} GB "
              f"(workspace {info['workspace_bytes'] / GB:.2f} GB), GPU total {info['dev_total'] / GB:.0f} GB")
        for b in (0, B // 2 - 1, B - 1):
            flow = ctx.last_flow(b)
            chain = fo.run_chain(flow, smp[b])
            assert tuple(res[b]["foe"]) == tuple(chain["foe"]), (b, tuple(res[b]["foe"]), chain["foe"])
            assert np.array_equal(mf[b], chain["fixed"]) and np.array_equal(md[b], chain["total"]), b
            assert tuple(res[b]["box"]) == tuple(chain["box"]), b
            check_flow(flow, fb_oracle.calc(prev[b], nxt[b], par), b)
        for b in range(B):                                             # every slot: box == extents of its own fixed mask, mask not empty
            assert tuple(res[b]["box"]) == fo.simple_bounding_box(mf[b]), b
            assert mf[b].any(), b
        # pairs built from the same base pair with different shifts must differ (slot b holds pair b, not a neighbour's result)
        assert len({res[b]["foe"].tobytes() for b in range(0, B, 4)}) > B // 8
        sums = (res.tobytes(), zlib.crc32(mf.view(np.uint8)), zlib.crc32(md.view(np.uint8)),
                [zlib.crc32(ctx.last_flow(b)) for b in (0, 1, B // 2, B - 1)])
        del mf, md
        res2, mf2, md2 = run()                                         # the timed loop repeats the call: deterministic, all pairs
        assert (res2.tobytes(), zlib.crc32(mf2.view(np.uint8)), zlib.crc32(md2.view(np.uint8)),
                [zlib.crc32(ctx.last_flow(b)) for b in (0, 1, B // 2, B - 1)]) == sums
        for d in (d_prev, d_next, d_smp, d_res, d_mf, d_md):
            d.free()


def test_c4_global_batch_1080p_512_pairs_on_one_gpu(mav, fb_oracle):
    soak(1920, 1080, 512, 1, fb_oracle, "C4 global batch, 1920x1080")


def test_c5_global_batch_4k_five_layers_128_pairs_on_one_gpu(mav, fb_oracle):
    soak(3840, 2160, 128, 5, fb_oracle, "C5 global batch, 3840x2160 / 5 layers")
