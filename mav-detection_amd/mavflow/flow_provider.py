"""The reference's flow seam -- Dataset.get_flow_uv(i) -> float32 (H, W, 2) (/root/reference/src/datasets/dataset.py:205-212,
called at src/processor.py:287,305) -- as two small providers a dataset object can delegate to:

    FloFlowProvider         what the reference does today: read `{img_path}/output/inference/run.epoch-0-flow-field/{i:06d}.flo`
                            (FlowNet2's output directory) with utils.read_flow.
    FarnebackFlowProvider   the same call answered by libmavflow's Farneback on the GPU from the sequence's gray frames, optionally
                            writing each field back as a .flo file in the reference's layout so that every other consumer of
                            those files (validator, plots) keeps working unchanged.

Flow i is the field from frame i to frame i + 1.  Errors follow the reference: a missing file is an OSError, a bad tag an
AssertionError (utils.py:217).
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import numpy as np

from . import utils


def flo_path(img_path: str, i: int) -> str:
    """dataset.py:211"""
    return f"{img_path}/output/inference/run.epoch-0-flow-field/{i:06d}.flo"


class FloFlowProvider:
    def __init__(self, img_path: str) -> None:
        self.img_path = img_path

    def get_flow_uv(self, i: int) -> np.ndarray:
        return utils.read_flow(flo_path(self.img_path, i))


class FarnebackFlowProvider:
    """get_gray(i) -> u8 (H, W) frame i (BGR frames are converted on the GPU).  window = 1: one pair per call, exactly the call
    shape of src/farneback.py:76-80.  window = n > 1 (with n_frames = the length of the video): a call for flow i that is not in
    the cache computes flows i .. i + n - 1 in one go from the n + 1 frames i .. i + n as a frame SEQUENCE -- one upload, every
    frame blurred and expanded once (Context.farneback_sequence) -- so the reference's frame-by-frame loop runs batched without
    changing; the fields are bit-identical to the one-pair calls."""

    def __init__(self, get_gray: Callable[[int], np.ndarray], width: int, height: int, img_path: Optional[str] = None,
                 write_flo: bool = False, window: int = 1, n_frames: Optional[int] = None, on_device: bool = False,
                 lanes: Optional[int] = None) -> None:
        """on_device (window = 1): get_flow_uv returns a pipeline.DeviceArray -- the field stays on the GPU until somebody reads it,
        which Processor.run_detection never does (BGR frames are converted there too); off by default: a host float32 array.
        lanes: contexts the on-device seam takes in turn (None: pipeline.auto_lanes -- 4 up to ~720p, 3 at 1080p, 1 beyond)."""
        from . import _lib
        self.get_gray, self.img_path, self.write_flo = get_gray, img_path, write_flo
        if window < 1 or (window > 1 and n_frames is None):
            raise ValueError("window must be >= 1, and a window > 1 needs n_frames")
        self.window, self.n_frames = int(window), n_frames
        self.ctx = _lib.Context(width, height, self.window)
        self._cache: dict = {}
        self._stage = None
        self._lane_ctxs = []
        if on_device:
            if self.window != 1:
                raise ValueError("on_device needs window = 1")
            from . import pipeline
            n = lanes or pipeline.auto_lanes(width, height, 1)
            self._lane_ctxs = [_lib.Context(width, height, 1) for _ in range(n - 1)]
            self._stage = pipeline.LanedFlowStage([self.ctx] + self._lane_ctxs)
        if write_flo and not img_path:
            raise ValueError("write_flo needs img_path")

    @classmethod
    def from_png_sequence(cls, img_path: str, img_format: str = "image_%05d.png", **kw) -> "FarnebackFlowProvider":
        """The reference's frame layout (src/datasets/dataset.py:26,38: `{img_path}/image_%05d.png`) as the source of the frames: frame i
        = cv2.imread of file i (frame_source.imread), converted BGR -> gray on the GPU."""
        from . import frame_source
        pattern = f"{img_path}/{img_format}"
        first = frame_source.imread(pattern % 0)
        if first is None:
            raise OSError(f"cannot read {pattern % 0}")

        def get(i: int) -> np.ndarray:
            f = frame_source.imread(pattern % i)
            if f is None:
                raise OSError(f"cannot read {pattern % i}")
            return f
        kw.setdefault("img_path", img_path)
        return cls(get, first.shape[1], first.shape[0], **kw)

    def _gray(self, i: int) -> np.ndarray:
        f = np.asarray(self.get_gray(i))
        return self.ctx.bgr2gray(f)[0] if f.ndim == 3 else np.ascontiguousarray(f, np.uint8)

    def _compute(self, i: int) -> np.ndarray:
        if self._stage is not None:
            return self._stage.flow_of(self.get_gray(i), self.get_gray(i + 1))
        if self.window == 1:
            return self.ctx.farneback(self._gray(i), self._gray(i + 1))[0]
        if i not in self._cache:
            n = max(1, min(self.window, self.n_frames - 1 - i))
            flows = self.ctx.farneback_sequence(np.stack([self._gray(j) for j in range(i, i + n + 1)]))
            self._cache = {i + k: flows[k] for k in range(n)}         # views of one (pinned) result block
        return self._cache[i]

    def get_flow_uv(self, i: int) -> np.ndarray:
        flow = self._compute(i)
        if self.write_flo:
            path = flo_path(self.img_path, i)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            utils.write_flow(path, np.asarray(flow))
        return flow

    def release(self) -> None:
        if self._stage is not None:
            self._stage.close()
        for c in self._lane_ctxs:
            c.close()
        self.ctx.close()
