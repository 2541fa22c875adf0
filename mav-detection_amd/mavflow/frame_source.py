"""Frame decode in front of the path: the reference reads its frames with cv2.VideoCapture over `{img_path}/image_%05d.png`
(/root/reference/src/datasets/dataset.py:26,38,57,223-230) and hands them to Farneback as BGR u8 arrays (src/farneback.py:17-21,73-74).
cv2 is not part of this build; this module is that capture for PNG sequences:

    decode_png(bytes) -> u8 array        chunk parsing + zlib inflate (Python stdlib), un-filtering in libmavflow (mav_png_unfilter:
                                         two of PNG's five filters are per-pixel recurrences).  Every colour type (gray, gray + alpha,
                                         RGB, RGBA, palette) at every legal bit depth (1 / 2 / 4 / 8 / 16), non-interlaced and Adam7
                                         interlaced; 16-bit samples are narrowed to their high byte, which is what cv2.imread's
                                         default flag does (libpng's png_set_strip_16).
    imread(path) -> BGR u8 (H, W, 3)     what cv2.imread(path) (IMREAD_COLOR) returns: gray replicated, alpha dropped, palette expanded.
    PngSequenceCapture(pattern)          cv2.VideoCapture(pattern)'s read() / get(3|4|7) / isOpened() / release() for an image sequence.

The decoder is pinned by PIL-decoded fixtures generated in the build container (tools/gen_png_fixtures.py,
tests/golden/png_frames.npz), 16-bit and interlaced files included (round 6).
"""
from __future__ import annotations

import os
import struct
import zlib
from typing import Optional, Tuple

import numpy as np

from . import _lib

PNG_MAGIC = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}          # colour type -> samples per pixel


def decode_png(data: bytes) -> Tuple[np.ndarray, int]:
    """(pixels, colour type): pixels u8 (H, W) for gray / (H, W, 2) gray + alpha / (H, W, 3) RGB / (H, W, 4) RGBA; palette images
    come back expanded to RGB (or RGBA when the file has a tRNS chunk), colour type 2 / 6."""
    if data[:8] != PNG_MAGIC:
        raise ValueError("not a PNG file (bad signature)")
    pos = 8
    ihdr = None
    idat = []
    plte = trns = None
    while pos + 8 <= len(data):
        (length,), kind = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        if len(body) != length or pos + 12 + length > len(data):
            raise ValueError("truncated PNG chunk")
        (crc,) = struct.unpack(">I", data[pos + 8 + length:pos + 12 + length])
        if zlib.crc32(kind + body) & 0xFFFFFFFF != crc:
            raise ValueError(f"PNG chunk {kind!r}: CRC mismatch")
        pos += 12 + length
        if kind == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = np.frombuffer(body, np.uint8)
        elif kind == b"IEND":
            break
    if ihdr is None or not idat:
        raise ValueError("PNG without IHDR / IDAT")
    W, H, depth, ctype, comp, filt, interlace = ihdr
    if ctype not in _CHANNELS or comp != 0 or filt != 0:
        raise ValueError(f"unsupported PNG header (colour type {ctype}, compression {comp}, filter method {filt})")
    if interlace not in (0, 1):
        raise ValueError(f"unknown PNG interlace method {interlace}")
    if depth not in (1, 2, 4, 8, 16) or (depth < 8 and ctype not in (0, 3)) or (depth == 16 and ctype == 3):
        raise ValueError(f"invalid PNG bit depth {depth} for colour type {ctype}")
    ch = _CHANNELS[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    lib = _lib.load()

    def samples(off, pw, ph):
        """one (sub-)image of pw x ph pixels starting at byte `off` of the inflated stream -> ((ph, pw * ch) u8 sample values, next offset)"""
        stride = (pw * ch * depth + 7) // 8
        n = ph * (stride + 1)
        if off + n > len(raw):
            raise ValueError(f"PNG image data holds {len(raw)} bytes, more expected")
        rows = np.empty((ph, stride), np.uint8)
        _lib.check(lib.mav_png_unfilter(raw[off:off + n].ctypes.data, ph, stride, max(1, ch * depth // 8), rows.ctypes.data))
        if depth == 8:
            v = rows
        elif depth == 16:                            # big-endian 16-bit samples -> their high byte (png_set_strip_16)
            v = np.ascontiguousarray(rows.reshape(ph, pw * ch, 2)[..., 0])
        else:                                        # samples packed most significant bit first
            bits = np.unpackbits(rows, axis=1)[:, :pw * depth].reshape(ph, pw, depth)
            v = np.zeros((ph, pw), np.uint8)
            for b in range(depth):
                v = (v << 1) | bits[..., b]
        return v, off + n

    if interlace == 0:
        out, end = samples(0, W, H)
    else:                                            # Adam7: seven reduced images, each filtered on its own, scattered onto the 8 x 8 lattice
        out = np.empty((H, W * ch), np.uint8)
        grid = out.reshape(H, W, ch)
        end = 0
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            pw, ph = (W - x0 + dx - 1) // dx, (H - y0 + dy - 1) // dy
            if pw <= 0 or ph <= 0:
                continue                             # an empty pass has no bytes at all, not even filter bytes
            v, end = samples(end, pw, ph)
            grid[y0::dy, x0::dx] = v.reshape(ph, pw, ch)
    if end != len(raw):
        raise ValueError(f"PNG image data holds {len(raw)} bytes, {end} expected")
    if depth < 8 and ctype == 0:                     # gray levels scale to 0 .. 255
        out = (out * (255 // ((1 << depth) - 1))).astype(np.uint8)
    out = out.reshape(H, W) if ch == 1 else out.reshape(H, W, ch)
    if ctype == 3:
        if plte is None:
            raise ValueError("palette PNG without PLTE chunk")
        if int(out.max(initial=0)) >= len(plte):
            raise ValueError("palette index out of range")
        rgb = plte[out]
        if trns is not None:
            alpha = np.full(len(plte), 255, np.uint8)
            alpha[:len(trns)] = trns[:len(plte)]
            return np.dstack([rgb, alpha[out]]), 6
        return rgb, 2
    return out, ctype


def imread(path: str) -> Optional[np.ndarray]:
    """cv2.imread(path): BGR u8 (H, W, 3), None when the file does not exist or cannot be read as a PNG.  (Gray replicated into three
    channels, alpha dropped -- IMREAD_COLOR, the flag the reference's calls default to.)"""
    try:
        with open(path, "rb") as f:
            data = f.read()
        px, ctype = decode_png(data)
    except (OSError, ValueError, zlib.error):
        return None
    if ctype in (0, 4):
        g = px if ctype == 0 else px[..., 0]
        return np.repeat(g[..., None], 3, axis=2)
    return np.ascontiguousarray(px[..., 2::-1])      # RGB(A) -> BGR


class PngSequenceCapture:
    """cv2.VideoCapture('.../image_%05d.png') for a PNG sequence: frames are the files pattern % k for k = start, start + 1, ... until
    one is missing (OpenCV's image-sequence capture starts at the first existing index among 0 and 1; here: `start`, default the same
    probe).  read() -> (ok, BGR frame) like cv2; get(3) / get(4) / get(7) = width / height / frame count; set(1, k) seeks."""

    def __init__(self, pattern: str, start: Optional[int] = None) -> None:
        self.pattern = pattern
        if start is None:
            start = 0 if os.path.exists(pattern % 0) else 1
        self.start = self.pos = start
        first = imread(pattern % start)
        self._opened = first is not None
        self._size = (first.shape[1], first.shape[0]) if self._opened else (0, 0)
        self._count = None

    def isOpened(self) -> bool:
        return self._opened

    def read(self):
        if not self._opened:
            return False, None
        frame = imread(self.pattern % self.pos)
        if frame is None:
            return False, None
        self.pos += 1
        return True, frame

    def get(self, prop: int) -> float:
        if prop == 3:
            return float(self._size[0])
        if prop == 4:
            return float(self._size[1])
        if prop == 7:                                # CAP_PROP_FRAME_COUNT
            if self._count is None:
                k = self.start
                while os.path.exists(self.pattern % k):
                    k += 1
                self._count = k - self.start
            return float(self._count)
        if prop == 1:                                # CAP_PROP_POS_FRAMES
            return float(self.pos - self.start)
        return 0.0

    def set(self, prop: int, value: float) -> bool:
        if prop == 1:
            self.pos = self.start + int(value)
            return True
        return False

    def release(self) -> None:
        self._opened = False
