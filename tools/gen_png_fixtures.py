"""Fixtures for mavflow/frame_source.py: small PNG files and what PIL decodes them to.  Run in the build container (PIL is importable
there; it is not needed to run the tests):

    python tools/gen_png_fixtures.py            -> tests/golden/png_frames.npz

Per case: `png_<name>` = the file's bytes (uint8), `bgr_<name>` = cv2.imread's view of it derived from PIL's decode (RGB -> BGR, gray
replicated, alpha dropped).  Files come from two encoders: PIL's own (adaptive filter choice) and the minimal encoder below, which
forces filter type (row + k) % 5 on every row so that all five filters -- and their first-row / first-pixel edge cases -- occur in
every colour type and bit depth.  Round 6: 16-bit samples (cv2.imread's default flag narrows them to the high byte, libpng's
png_set_strip_16 -- which is also what PIL's "RGB" view of a 16-bit RGB file holds; for 16-bit GRAY files PIL keeps all sixteen bits,
so the expectation is its array >> 8) and Adam7-interlaced files (PIL reads them; the encoder below writes the seven passes, every pass
filtered on its own, sizes down to 1 x 1 so that empty passes occur).  Nothing here needs cv2."""
import io
import os
import struct
import sys
import zlib

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def chunk(kind, body):
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def filter_rows(rows_bytes, bpp, first_filter):
    """(H, stride) u8 of packed samples -> the filtered byte stream.  Filter type of row y = (y + first_filter) % 5."""
    H, stride = rows_bytes.shape
    raw = bytearray()
    prev = np.zeros(stride, np.int64)
    for y in range(H):
        cur = rows_bytes[y].astype(np.int64)
        ft = (y + first_filter) % 5
        out = np.zeros(stride, np.int64)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, paeth(a, b, c))[ft]
            out[i] = (cur[i] - pred) & 255
        raw.append(ft)
        raw += bytes(out.astype(np.uint8))
        prev = cur
    return bytes(raw)


ADAM7 = ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2))


def encode_adam7(samples, depth, ctype, first_filter=0, extra=b""):
    """samples: (H, W, ch) integer sample values (< 2^depth) -> an Adam7-interlaced PNG file."""
    H, W, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    raw = b""
    for k, (x0, y0, dx, dy) in enumerate(ADAM7):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        ph, pw = sub.shape[:2]
        if depth == 16:
            rows = sub.astype(">u2").reshape(ph, -1).view(np.uint8)
        elif depth == 8:
            rows = sub.astype(np.uint8).reshape(ph, -1)
        else:
            rows = pack_bits(sub[..., 0].astype(np.uint8), depth)
        raw += filter_rows(np.ascontiguousarray(rows), bpp, first_filter + k)
    ihdr = struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 1)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + extra + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def encode(rows_bytes, W, H, depth, ctype, bpp, first_filter=0, extra=b"", idat_split=1):
    """rows_bytes: (H, stride) u8 of packed samples.  Filter type of row y = (y + first_filter) % 5."""
    stride = rows_bytes.shape[1]
    raw = bytearray()
    prev = np.zeros(stride, np.int64)
    for y in range(H):
        cur = rows_bytes[y].astype(np.int64)
        ft = (y + first_filter) % 5
        out = np.zeros(stride, np.int64)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, paeth(a, b, c))[ft]
            out[i] = (cur[i] - pred) & 255
        raw.append(ft)
        raw += bytes(out.astype(np.uint8))
        prev = cur
    comp = zlib.compress(bytes(raw), 6)
    n = max(1, len(comp) // idat_split)
    parts = [comp[i:i + n] for i in range(0, len(comp), n)]
    ihdr = struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + extra + b"".join(chunk(b"IDAT", p) for p in parts) + chunk(b"IEND", b"")


def pack_bits(vals, depth):
    H, W = vals.shape
    bits = ((vals[..., None] >> np.arange(depth - 1, -1, -1)) & 1).astype(np.uint8).reshape(H, W * depth)
    return np.packbits(bits, axis=1)


def as_bgr(png_bytes):
    im = Image.open(io.BytesIO(png_bytes))
    im.load()
    if im.mode.startswith("I;16") or im.mode == "I":       # 16-bit gray: PIL keeps sixteen bits (and its own convert() CLAMPS them);
        hi = (np.asarray(im).astype(np.uint32) >> 8).astype(np.uint8)      # cv2.imread's default flag strips to the high byte
        return np.repeat(hi[..., None], 3, axis=2)
    rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[..., ::-1])


def main():
    rng = np.random.default_rng(5)
    cases = {}

    def smooth(H, W, ch):
        y, x = np.mgrid[0:H, 0:W]
        base = np.stack([(3 * x + 2 * y + 40 * c) % 256 for c in range(ch)], axis=-1)
        noise = rng.integers(-6, 7, (H, W, ch))
        return np.clip(base + noise, 0, 255).astype(np.uint8)

    g = smooth(37, 48, 1)[..., 0]
    cases["gray8_forced"] = encode(g, 48, 37, 8, 0, 1, first_filter=0)
    rgb = smooth(29, 33, 3)
    cases["rgb8_forced"] = encode(rgb.reshape(29, -1), 33, 29, 8, 2, 3, first_filter=3, idat_split=3)      # first row Average, IDAT in 3 chunks
    rgba = smooth(17, 20, 4)
    cases["rgba8_forced"] = encode(rgba.reshape(17, -1), 20, 17, 8, 6, 4, first_filter=4)                 # first row Paeth
    la = smooth(11, 13, 2)
    cases["graya8_forced"] = encode(la.reshape(11, -1), 13, 11, 8, 4, 2, first_filter=1)
    for depth in (1, 2, 4):
        v = rng.integers(0, 1 << depth, (15, 21)).astype(np.uint8)
        cases[f"gray{depth}_forced"] = encode(pack_bits(v, depth), 21, 15, depth, 0, 1, first_filter=depth)
    pal = rng.integers(0, 256, (16, 3)).astype(np.uint8)
    idx = rng.integers(0, 16, (14, 19)).astype(np.uint8)
    cases["pal4_forced"] = encode(pack_bits(idx, 4), 19, 14, 4, 3, 1, first_filter=2, extra=chunk(b"PLTE", pal.tobytes()))
    pal8 = rng.integers(0, 256, (200, 3)).astype(np.uint8)
    idx8 = rng.integers(0, 200, (12, 18)).astype(np.uint8)
    cases["pal8_trns_forced"] = encode(idx8, 18, 12, 8, 3, 1, first_filter=0,
                                       extra=chunk(b"PLTE", pal8.tobytes()) + chunk(b"tRNS", rng.integers(0, 256, 50).astype(np.uint8).tobytes()))
    # PIL's own encoder: adaptive filter choice, its own chunk layout
    for name, arr, mode in (("rgb8_pil", smooth(48, 64, 3), "RGB"), ("gray8_pil", smooth(40, 56, 1)[..., 0], "L"), ("rgba8_pil", smooth(24, 32, 4), "RGBA")):
        buf = io.BytesIO()
        Image.fromarray(arr, mode).save(buf, format="PNG", optimize=True)
        cases[name] = buf.getvalue()
    buf = io.BytesIO()
    Image.fromarray(smooth(30, 40, 3), "RGB").quantize(32).save(buf, format="PNG")
    cases["pal_pil"] = buf.getvalue()
    # 16-bit samples, non-interlaced: gray, RGB, RGBA, gray + alpha
    g16 = rng.integers(0, 65536, (13, 17)).astype(">u2")
    cases["gray16_forced"] = encode(g16.reshape(13, -1).view(np.uint8), 17, 13, 16, 0, 2, first_filter=2)
    rgb16 = rng.integers(0, 65536, (11, 14, 3)).astype(">u2")
    cases["rgb16_forced"] = encode(rgb16.reshape(11, -1).view(np.uint8), 14, 11, 16, 2, 6, first_filter=4)
    rgba16 = rng.integers(0, 65536, (9, 10, 4)).astype(">u2")
    cases["rgba16_forced"] = encode(rgba16.reshape(9, -1).view(np.uint8), 10, 9, 16, 6, 8, first_filter=1)
    la16 = rng.integers(0, 65536, (7, 9, 2)).astype(">u2")
    cases["graya16_forced"] = encode(la16.reshape(7, -1).view(np.uint8), 9, 7, 16, 4, 4, first_filter=3)
    # Adam7: every colour type, depths 1 .. 16, sizes with empty passes (1 x 1, 2 x 2, 3 x 5), one larger than two lattice cells
    cases["adam7_gray8"] = encode_adam7(smooth(21, 19, 1), 8, 0, first_filter=1)
    cases["adam7_rgb8"] = encode_adam7(smooth(18, 23, 3), 8, 2, first_filter=3)
    cases["adam7_rgba8"] = encode_adam7(smooth(9, 12, 4), 8, 6, first_filter=0)
    cases["adam7_graya8"] = encode_adam7(smooth(10, 7, 2), 8, 4, first_filter=2)
    for depth in (1, 2, 4):
        cases[f"adam7_gray{depth}"] = encode_adam7(rng.integers(0, 1 << depth, (11, 13, 1)), depth, 0, first_filter=depth)
    cases["adam7_pal4"] = encode_adam7(rng.integers(0, 16, (12, 9, 1)), 4, 3, first_filter=4, extra=chunk(b"PLTE", pal.tobytes()))
    cases["adam7_rgb16"] = encode_adam7(rng.integers(0, 65536, (8, 9, 3)), 16, 2, first_filter=2)
    cases["adam7_gray16"] = encode_adam7(rng.integers(0, 65536, (9, 8, 1)), 16, 0, first_filter=0)
    for (h, w) in ((1, 1), (2, 2), (3, 5), (5, 3), (1, 9), (9, 1)):
        cases[f"adam7_rgb8_{w}x{h}"] = encode_adam7(smooth(h, w, 3), 8, 2, first_filter=h + w)
    out = {}
    for name, data in cases.items():
        out["png_" + name] = np.frombuffer(data, np.uint8)
        out["bgr_" + name] = as_bgr(data)
    # a three-frame sequence at the hot path's smallest interesting size, for the capture / Farneback wiring test
    from importlib import import_module
    sys.path.insert(0, os.path.join(ROOT, "mav-detection_amd"))
    synth = import_module("mavflow.synth")
    seq = synth.make_sequence(96, 64, 3)
    for k in range(3):
        rgbk = np.repeat(seq[k][..., None], 3, axis=2).copy()
        rgbk[..., 0] = np.clip(rgbk[..., 0].astype(int) + 9, 0, 255)      # not a gray replica: BGR2GRAY must do real work
        buf = io.BytesIO()
        Image.fromarray(rgbk, "RGB").save(buf, format="PNG")
        out[f"png_seq{k}"] = np.frombuffer(buf.getvalue(), np.uint8)
        out[f"bgr_seq{k}"] = as_bgr(buf.getvalue())
    path = os.path.join(ROOT, "tests", "golden", "png_frames.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {len(cases) + 3} PNG files, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
