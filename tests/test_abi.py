"""CPU-side checks of the drop-in boundary: libmavflow.so loads, exports every symbol include/mavflow.h declares,
has the struct layouts the header states, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mavflow.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mav_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(mav):
    from mavflow import _lib
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"libmavflow.so does not export {s}"
    assert sorted(_lib.EXPORTS) == syms


def test_struct_layouts(mav):
    from mavflow import _lib
    assert C.sizeof(_lib.Result) == 32 and _lib.RESULT_DTYPE.itemsize == 32
    assert _lib.RESULT_DTYPE.fields["foe"][1] == 16
    fb = _lib.fb_defaults()
    assert (fb.pyr_scale, fb.levels, fb.winsize, fb.iterations, fb.poly_n, fb.poly_sigma, fb.flags) == (0.4, 1, 12, 10, 8, 1.2, 0)
    fo = _lib.foe_defaults()
    assert (fo.n_pairs, fo.mag_threshold, fo.ransac_threshold) == (1000, 2.5, 30.0)
    th = _lib.thr_defaults()
    assert (th.fixed_deg, th.fixed_min_mag, th.dyn_min_mag, th.dyn_a, th.dyn_b, th.dyn_c) == (15.0, 1.0, 0.5, 0.25, 0.5, 8.0)


def test_no_cpu_fallback(mav):
    """Without a GPU the product path must fail loudly, never compute on the host."""
    from mavflow import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.MavflowError):
        _lib.Context(64, 48, 1)


def test_bad_arguments_are_value_errors(mav):
    from mavflow import _lib
    fb = _lib.fb_defaults()
    fb.pyr_scale = 1.0                       # cv2 raises for pyr_scale >= 1
    with pytest.raises(ValueError):
        _lib.Context(64, 48, 1, fb)
    fb = _lib.fb_defaults()
    fb.flags = 256                           # OPTFLOW_FARNEBACK_GAUSSIAN: not implemented, says so
    with pytest.raises(ValueError):
        _lib.Context(64, 48, 1, fb)
    with pytest.raises(ValueError):
        _lib.Context(0, 48, 1)
    fb = _lib.fb_defaults()
    fb.winsize = 62                          # general sweep kernel: 5 x (32 + 62)^2 floats = 176 KB of LDS > the CU's 160 KB
    with pytest.raises(ValueError, match="LDS"):
        _lib.Context(640, 480, 1, fb)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mav-detection_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "farneback_oracle" not in src and "libfboracle" not in src, f


def test_size_bound_is_an_argument_error(mav):
    """max_batch * W * H above 2^30 pixels (include/mavflow.h: MAV_MAX_BATCH_PIXELS) is refused before anything touches a device."""
    from mavflow import _lib
    h = C.c_void_p()
    lib = _lib.load()
    for (W, H, B) in ((1920, 1080, 518), (3840, 2160, 130), (64, 48, 65536)):
        assert lib.mav_create(C.byref(h), 0, W, H, B, None) == _lib.MAV_ERR_ARG, (W, H, B)
        assert b"bound" in lib.mav_last_error()
    txt = open(os.path.join(ROOT, "include", "mavflow.h")).read()
    assert "MAV_MAX_BATCH_PIXELS ((size_t)1 << 30)" in txt
    assert 1920 * 1080 * 512 <= 1 << 30 and 3840 * 2160 * 128 <= 1 << 30       # BASELINE configs 4 and 5 are inside it


def test_no_environment_override_of_the_library_path(mav, monkeypatch):
    from mavflow import _lib
    src = open(_lib.__file__).read()
    assert "os.environ" not in src and "getenv" not in src
    assert _lib.SO_PATH.endswith(os.path.join("mavflow", "libmavflow.so"))


def test_frame_step_struct_layout_matches_the_header(mav, tmp_path):
    """mav_frame_step / mav_gather (include/mavflow.h) against their ctypes mirrors, field by field: offsets from a C program compiled
    against the header here and now."""
    import shutil
    import subprocess
    from mavflow import _lib
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mavflow.h"', "int main(void) {"]
    for cname, cls in (("mav_frame_step", _lib.FrameStep), ("mav_gather", _lib.Gather)):
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for name, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{name} %zu\\n", offsetof({cname}, {name}));')
    lines.append("  return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call([cc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in (("mav_frame_step", _lib.FrameStep), ("mav_gather", _lib.Gather)):
        assert int(got[cname]) == C.sizeof(cls)
        for name, _ in cls._fields_:
            assert int(got[f"{cname}.{name}"]) == getattr(cls, name).offset, (cname, name)
    assert _lib.STEP_MAX_GATHER == 4 and (_lib.GATHER_ORDERED, _lib.GATHER_SOURCES_HELD) == (1, 2)
