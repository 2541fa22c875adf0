"""FocusOfExpansion -- the dense FoE fit and the per-pixel radial residual on libmavflow, behind the reference's
call signatures (/root/reference/src/focus_of_expansion.py:13-86,150-184).

RNG ownership: get_FOE_dense draws its 2N sample coordinates from the global legacy numpy stream exactly as the
reference does (rows first, then columns, :70-71) and the constructor consumes the same draws (:24,26), so a seeded
run visits the same pixels.  The GPU never generates samples."""
from __future__ import annotations

from typing import Tuple

import numpy as np

from . import _lib, im_helpers


class FocusOfExpansion:
    def __init__(self, lucas_kanade) -> None:
        """`lucas_kanade`: any object with .total_num_corners and .old_frame (H, W[, 3]) -- see detector.LucasKanade."""
        self.lucas_kanade = lucas_kanade
        self.time = 0
        self.roll_back = 20
        self.num_features = 0
        self.enable_plots = False
        self.max_flow = 0.0                       # maximum phi in the image (degrees)
        self.radial_threshold = np.cos(np.deg2rad(15))
        self.magnitude_threshold = 2.5
        self.ransac_threshold = 30.0              # pixels
        n = int(lucas_kanade.total_num_corners)
        self.color = np.random.randint(0, 255, (n, 3))
        self.trace = np.zeros((n, 2000), dtype=np.int32)
        self.random_lines = np.random.randint(0, n, n)
        self.flow_height, self.flow_width = lucas_kanade.old_frame.shape[0], lucas_kanade.old_frame.shape[1]

    def _ctx(self, flow: np.ndarray) -> "_lib.Context":
        H, W = flow.shape[:2]
        return im_helpers._ctx(W, H)

    def _foe_params(self, n_pairs: int = 1000) -> "_lib.FoeParams":
        p = _lib.foe_defaults()
        p.n_pairs = n_pairs
        p.mag_threshold = float(self.magnitude_threshold)
        p.ransac_threshold = float(self.ransac_threshold)
        return p

    def ransac(self, estimates: np.ndarray) -> Tuple[float, float]:
        """First estimate with the strictly largest number of others within ransac_threshold; (0.0, 0.0) if none has any."""
        return im_helpers._ctx(self.flow_width, self.flow_height).ransac(estimates, self.ransac_threshold)

    def get_FOE_dense(self, flow_uv: np.ndarray) -> Tuple[float, float]:
        """FoE from N = 1000 random flow-line intersections + the RANSAC vote.  The arithmetic follows the array's dtype as
        numpy's does: a float32 field (frame index 0, which derotate hands back untouched) has its |flow2| gate evaluated in
        float32, a float64 field in double."""
        N = 1000
        rand1 = np.zeros((N * 2, 2), dtype=np.uint32)
        rand1[..., 0] = np.random.randint(0, flow_uv.shape[0], N * 2)
        rand1[..., 1] = np.random.randint(0, flow_uv.shape[1], N * 2)
        foe = self._ctx(flow_uv).foe_dense(flow_uv, rand1, self._foe_params(N))[0]
        return (float(foe[0]), float(foe[1]))

    def get_phi(self, derotated_flow_uv: np.ndarray, FoE: Tuple[float, float]) -> np.ndarray:
        """Angle (degrees) between each flow vector and the ray from the FoE through its pixel; max goes to .max_flow.
        float32 flow in -> float32 arithmetic and float32 phi out (zeros_like in the reference), float64 otherwise."""
        if FoE[0] is np.nan:                      # identity test, as the reference (:160)
            return np.zeros(0)
        phi, _, _, mx = self._ctx(derotated_flow_uv).phi_mask(derotated_flow_uv, FoE)
        self.max_flow = mx[0]
        return phi[0]

    def get_masks(self, derotated_flow_uv: np.ndarray, FoE: Tuple[float, float], sky_mask=None, params=None):
        """The threshold block of processor.py:333-341 -> (estimate_fixed, total_mask); phi is not materialised."""
        _, fixed, total, _ = self._ctx(derotated_flow_uv).phi_mask(derotated_flow_uv, FoE, sky=sky_mask, params=params, want_phi=False)
        return fixed[0], total[0]
