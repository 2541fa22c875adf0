"""cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) on uint8, restated in numpy.  TEST INFRASTRUCTURE ONLY (checker of mav_bgr2gray).

Call sites in the reference: /root/reference/src/farneback.py:21,74.  The arithmetic lives in OpenCV (opencv-python, unpinned,
requirements.txt:4), which is absent from the reference tree and from this image: PARITY UNPINNED.  Restated from OpenCV 4.x
imgproc/src/color_rgb.simd.hpp (RGB2Gray<uchar>): fixed point with 14 fractional bits,
    gray = (B * 1868 + G * 9617 + R * 4899 + (1 << 13)) >> 14          (0.114, 0.587, 0.299 scaled by 2^14),
the same formula on every SIMD path.  What bounds it without cv2: the five known answers in tests (pure blue / green / red ->
29 / 150 / 76, white -> 255) and the weights summing to 2^14 exactly (gray stays gray).
"""
import numpy as np

B2Y, G2Y, R2Y, SHIFT = 1868, 9617, 4899, 14
assert B2Y + G2Y + R2Y == 1 << SHIFT


def bgr_to_gray(img: np.ndarray) -> np.ndarray:
    a = np.asarray(img)
    if a.ndim == 2:
        return np.ascontiguousarray(a, np.uint8)
    b, g, r = (a[..., i].astype(np.uint32) for i in range(3))
    return ((b * B2Y + g * G2Y + r * R2Y + (1 << (SHIFT - 1))) >> SHIFT).astype(np.uint8)
