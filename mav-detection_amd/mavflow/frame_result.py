"""FrameResult -- the per-frame output record, unchanged from /root/reference/src/frame_result.py:4-17
(12 plain attributes; no box field: the reference never stores one)."""
from __future__ import annotations


class FrameResult:
    FIELDS = ("time", "tpr", "fpr", "tpr_fixed", "fpr_fixed", "sky_tpr", "sky_fpr", "drone_size_pixels",
              "drone_flow_pixels", "foe_dense", "foe_gt", "center_phi")

    def __init__(self) -> None:
        for name in self.FIELDS:
            setattr(self, name, (0.0, 0.0) if name in ("drone_flow_pixels", "foe_dense", "foe_gt") else 0.0)

    def __repr__(self) -> str:
        return "FrameResult(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in self.FIELDS) + ")"
