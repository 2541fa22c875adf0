"""Frame decode in front of the path and the per-frame JSON behind it, CPU only.

The PNG decoder (mavflow/frame_source.py; un-filtering in libmavflow's host function mav_png_unfilter, which needs no GPU) against
fixtures PIL decoded in the build container (tools/gen_png_fixtures.py -> tests/golden/png_frames.npz): every colour type, bit
depths 1 / 2 / 4 / 8 / 16, all five filters forced on every row position, PIL's own adaptive encoder, palette + tRNS, split IDAT,
Adam7-interlaced files of every colour type down to 1 x 1 pixels (empty passes).
The JSON writer (Processor._store = src/processor.py:83-84) against the text the reference's own utils.get_json produced."""
import json
import logging
import os
import struct
import zlib

import numpy as np
import pytest

from mavflow import frame_source as fs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def png():
    return np.load(os.path.join(GOLDEN, "png_frames.npz"), allow_pickle=False)


def _names(z):
    return [k[4:] for k in z.files if k.startswith("png_")]


def test_every_fixture_decodes_to_what_pil_saw(png, tmp_path):
    names = _names(png)
    assert len(names) >= 36 and sum(n.startswith("adam7_") for n in names) >= 16 and sum("16" in n for n in names) >= 6
    for name in names:
        path = tmp_path / f"{name}.png"
        path.write_bytes(png["png_" + name].tobytes())
        got = fs.imread(str(path))
        exp = png["bgr_" + name]
        assert got is not None and got.dtype == np.uint8 and got.shape == exp.shape, name
        assert np.array_equal(got, exp), name
        assert got.flags.c_contiguous


def test_decode_png_returns_the_file_s_own_channels(png):
    px, ct = fs.decode_png(png["png_rgba8_forced"].tobytes())
    assert ct == 6 and px.shape == (17, 20, 4)
    assert np.array_equal(px[..., 2::-1], png["bgr_rgba8_forced"])
    px, ct = fs.decode_png(png["png_gray8_forced"].tobytes())
    assert ct == 0 and px.shape == (37, 48)
    px, ct = fs.decode_png(png["png_pal8_trns_forced"].tobytes())
    assert ct == 6 and px.shape == (12, 18, 4)             # palette + tRNS expands to RGBA


def _rechunk(data: bytes, edit):
    """Apply edit(kind, body) -> body to every chunk and re-seal the CRCs."""
    out, pos = [data[:8]], 8
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        kind, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        body = edit(kind, body)
        out.append(struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF))
        pos += 12 + n
    return b"".join(out)


def test_malformed_and_unsupported_files(png, tmp_path):
    good = png["png_rgb8_forced"].tobytes()
    with pytest.raises(ValueError):
        fs.decode_png(b"JFIF" + good[4:])
    bad_crc = bytearray(good); bad_crc[40] ^= 1
    with pytest.raises(ValueError):
        fs.decode_png(bytes(bad_crc))
    with pytest.raises(ValueError):
        fs.decode_png(good[:len(good) // 2])
    # a header that lies about the layout of the data (interlace flag / bit depth flipped, data untouched): a pass runs into a byte
    # that is no filter type, or the byte count does not add up
    interlaced = _rechunk(good, lambda k, b: b[:12] + b"\x01" if k == b"IHDR" else b)
    with pytest.raises(ValueError):
        fs.decode_png(interlaced)
    deep = _rechunk(good, lambda k, b: b[:8] + b"\x10" + b[9:] if k == b"IHDR" else b)
    with pytest.raises(ValueError, match="bytes"):
        fs.decode_png(deep)
    for hdr in (b"\x02", b"\x07"):                           # interlace method 2 does not exist; neither does bit depth 7
        with pytest.raises(ValueError):
            fs.decode_png(_rechunk(good, lambda k, b, hdr=hdr: (b[:12] + hdr if hdr == b"\x02" else b[:8] + hdr + b[9:]) if k == b"IHDR" else b))
    pal16 = _rechunk(png["png_pal8_trns_forced"].tobytes(), lambda k, b: b[:8] + b"\x10" + b[9:] if k == b"IHDR" else b)
    with pytest.raises(ValueError, match="bit depth"):
        fs.decode_png(pal16)                                   # palette images have no 16-bit form
    # a filter-type byte outside 0 - 4 is refused by the library, and imread() maps every failure to cv2.imread's None
    def bad_filter(k, b):
        if k != b"IDAT":
            return b
        raw = bytearray(zlib.decompress(b)); raw[0] = 7
        return zlib.compress(bytes(raw))
    one_idat = png["png_gray8_forced"].tobytes()
    with pytest.raises(ValueError):
        fs.decode_png(_rechunk(one_idat, bad_filter))
    p = tmp_path / "broken.png"
    p.write_bytes(bytes(bad_crc))
    assert fs.imread(str(p)) is None and fs.imread(str(tmp_path / "missing.png")) is None


def test_png_sequence_capture_reads_like_cv2_videocapture(png, tmp_path):
    for k in range(3):
        (tmp_path / f"image_{k:05d}.png").write_bytes(png[f"png_seq{k}"].tobytes())
    cap = fs.PngSequenceCapture(str(tmp_path / "image_%05d.png"))
    assert cap.isOpened() and (cap.get(3), cap.get(4), cap.get(7)) == (96.0, 64.0, 3.0)
    frames = []
    while True:
        ok, f = cap.read()
        if not ok:
            assert f is None
            break
        frames.append(f)
    assert len(frames) == 3
    for k in range(3):
        assert np.array_equal(frames[k], png[f"bgr_seq{k}"])
    assert cap.set(1, 1) and np.array_equal(cap.read()[1], png["bgr_seq1"])
    cap.release()
    assert not cap.isOpened() and cap.read() == (False, None)
    assert not fs.PngSequenceCapture(str(tmp_path / "nothing_%05d.png")).isOpened()


def test_processor_writes_the_reference_s_json_file_per_frame(tmp_path):
    """src/processor.py:83-84: `{results_path}/image_{i:05d}.json` holding json.dumps(utils.get_json(result), indent=4, sort_keys=True);
    src/validator.py:135-153 reads those files back key by key."""
    from mavflow.frame_result import FrameResult
    from mavflow.processor import Processor, SyntheticDataset
    from mavflow.run_config import RunConfig
    io_golden = np.load(os.path.join(GOLDEN, "frame0_io.npz"), allow_pickle=False)
    ds = SyntheticDataset(64, 48, 3, results_path=str(tmp_path / "results"))
    p = Processor(RunConfig(logging.getLogger("t"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
    r = FrameResult()
    r.foe_dense = (np.float64(297.87096720308574), np.float64(222.21492686396977))
    r.foe_gt = (352.0, 216.0)
    r.center_phi = np.float64(-143.13010235415598)
    r.tpr_fixed, r.fpr_fixed = np.float64(0.75), np.float64(0.001953125)
    r.tpr, r.fpr = np.float64(1.0) / np.float64(3.0), np.float64("nan")
    r.sky_tpr, r.sky_fpr = (0.0, 0.0)
    r.drone_flow_pixels = (np.float32(6.0), np.float32(-3.0))
    r.drone_size_pixels = np.sum(np.ones((24, 24)) > 0)
    r.time = 4 * (1 / 30.0)
    p._store(4, r)
    path = tmp_path / "results" / "image_00004.json"
    assert path.read_text() == str(io_golden["json_text"])
    assert p.detection_results[4] is r and p.config.results[4] is r
    back = json.loads(path.read_text())                   # what validator.load_results() picks out
    for key in ("time", "tpr", "fpr", "tpr_fixed", "fpr_fixed", "sky_tpr", "sky_fpr", "foe_dense", "foe_gt", "drone_flow_pixels",
                "drone_size_pixels", "center_phi"):
        assert key in back
    # a Processor without a results_path writes nothing
    ds2 = SyntheticDataset(64, 48, 3)
    p2 = Processor(RunConfig(logging.getLogger("t"), ds2, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
    p2._store(0, r)
    assert p2.results_path is None and sorted(os.listdir(tmp_path)) == ["results"]
