"""Host-side mirror of the reference's interface (pure Python pieces): Rectangle, line_intersection, .flo I/O, enums,
FrameResult -- against fixtures produced by the reference's own code where one exists.  CPU only."""
import logging

import numpy as np
import pytest

from mavflow import utils
from mavflow.frame_result import FrameResult
from mavflow.run_config import RunConfig


def test_rectangle_matches_reference(golden):
    r1 = utils.Rectangle.from_points((7, 5), (19, 8))
    r2 = utils.Rectangle.from_center((15.0, 8.0), (10.0, 6.0))
    vals = np.array([*r1.topleft, *r1.size, *r1.get_center(), r1.get_area(), utils.Rectangle.calculate_iou(r1, r2),
                     utils.Rectangle((0, 0), (0, 0)).get_area()], dtype=np.float64)
    assert vals.tobytes() == golden["rect_vals"].tobytes()
    assert r1.to_yolo(np.array([160, 120])) == str(golden["rect_yolo"])
    assert r1.get_bottomright() == (19, 8) and r1.get_topleft_int_offset() == (7, 0)
    e = utils.Rectangle.from_box((-1, -1, -1, -1))
    assert e.topleft == (-1, -1) and e.size == (0, 0)


def test_line_intersection_matches_reference(golden):
    pts, fl, exp = golden["li_pts"], golden["li_flow"], golden["li_out"]
    for i in range(pts.shape[0]):
        c1, c2 = pts[i, 0], pts[i, 1]
        got = utils.line_intersection((c1, fl[i, 0] + c1), (c2, fl[i, 1] + c2))
        assert np.array(got, dtype=np.float64).tobytes() == exp[i].tobytes(), i
    assert utils.line_intersection(((0, 0), (1, 1)), ((0, 1), (1, 2))) == (False, False)


def test_flo_roundtrip_and_bad_tag(tmp_path):
    rng = np.random.default_rng(0)
    flow = rng.normal(0, 3, (37, 53, 2)).astype(np.float32)
    p = tmp_path / "a.flo"
    utils.write_flow(str(p), flow)
    raw = p.read_bytes()
    assert raw[:4] == np.array([202021.25], np.float32).tobytes()
    assert np.frombuffer(raw[4:12], np.int32).tolist() == [53, 37]
    assert np.array_equal(utils.read_flow(str(p)), flow)
    utils.write_flow(str(p), flow[..., 0], flow[..., 1])
    assert np.array_equal(utils.read_flow(str(p)), flow)
    (tmp_path / "bad.flo").write_bytes(b"\0" * 64)
    with pytest.raises(AssertionError):
        utils.read_flow(str(tmp_path / "bad.flo"))
    empty = np.zeros((0, 0, 2), np.float32)
    utils.write_flow(str(p), empty)
    assert utils.read_flow(str(p)).shape == (0, 0, 2)


def test_enums_keep_the_reference_surface(golden):
    from mavflow.detector import Detector
    assert [a.name for a in Detector.Algorithm] == list(golden["algo_names"])
    assert [a.value for a in Detector.Algorithm] == [(int(v),) for v in golden["algo_values"]]     # 1-tuples
    assert [m.name for m in RunConfig.Mode] == ["APPEARANCE_RGB", "FLOW_UV", "FLOW_RADIAL", "FLOW_FOE_YOLO", "FLOW_FOE_CLUSTERING"]
    assert RunConfig.Mode["FLOW_FOE_CLUSTERING"].value == (4,)
    assert str(RunConfig.Mode.FLOW_UV) == "FLOW_UV"
    assert [m.name for m in RunConfig.DatasetType] == ["MIDGARD", "SIMULATION", "EXPERIMENT", "VIS_DRONE"]


def test_run_config_and_frame_result():
    cfg = RunConfig(logging.getLogger("t"), object(), "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING")
    assert cfg.mode is RunConfig.Mode.FLOW_FOE_CLUSTERING and not cfg.uses_nn_for_detection()
    assert cfg.get_dataset_type("simulation") is RunConfig.DatasetType.SIMULATION
    with pytest.raises(ValueError):
        RunConfig(logging.getLogger("t"), object(), "", False, False, False, True, False, False, "NOPE")
    with pytest.raises(NotImplementedError):
        RunConfig(logging.getLogger("t"), "simulation", "", False, False, False, True, False, False, "FLOW_UV").get_dataset()
    r = FrameResult()
    assert list(vars(r)) == ["time", "tpr", "fpr", "tpr_fixed", "fpr_fixed", "sky_tpr", "sky_fpr", "drone_size_pixels",
                             "drone_flow_pixels", "foe_dense", "foe_gt", "center_phi"]
    assert r.foe_dense == (0.0, 0.0) and r.time == 0.0
    r.drone_size_pixels = np.int64(7)
    assert utils.get_json(vars(r))["drone_size_pixels"] == "7"        # numpy ints become strings, as in the reference


def test_gray_and_int_helpers():
    from oracle.gray_oracle import bgr_to_gray
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 20, 30]]], np.uint8)
    assert bgr_to_gray(px).tolist() == [[29, 150, 76, 255, 22]]
    from mavflow import im_helpers       # imports _lib lazily; to_int / to_rgb are pure numpy
    a = np.array([[0.0, 90.0, 180.0]])
    assert im_helpers.to_int(a, np.uint8, True, 180.0).tolist() == [[0, 128, 255]]
    assert im_helpers.to_rgb(a, 180.0).shape == (1, 3, 3)


@pytest.fixture(scope="module")
def io_golden():
    import os
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "frame0_io.npz"), allow_pickle=False)


def test_flo_bytes_and_arrays_match_the_reference(tmp_path, io_golden):
    """utils.write_flow / read_flow against the bytes and arrays the reference's own functions produced
    (/root/reference/src/utils.py:204-257; tools/gen_golden.py)."""
    flo = io_golden["flo_in"]
    p = tmp_path / "a.flo"
    utils.write_flow(str(p), flo)
    assert p.read_bytes() == io_golden["flo_bytes"].tobytes()
    utils.write_flow(str(p), flo[..., 0].astype(np.float64), flo[..., 1].astype(np.float64))        # separate u, v planes
    assert p.read_bytes() == io_golden["flo_bytes_uv"].tobytes()
    p.write_bytes(io_golden["flo_bytes"].tobytes())                    # a file the reference wrote
    back = utils.read_flow(str(p))
    assert back.dtype == np.dtype(str(io_golden["flo_read_dtype"])) and back.tobytes() == io_golden["flo_read"].tobytes()
    assert back.shape == io_golden["flo_read"].shape == (37, 53, 2)


def test_flo_flow_provider_reads_the_reference_layout(tmp_path, io_golden):
    """Dataset.get_flow_uv (/root/reference/src/datasets/dataset.py:205-212): same path layout, same array, same errors."""
    from mavflow.flow_provider import FloFlowProvider, flo_path
    img = tmp_path / "seq" / "images"
    path = flo_path(str(img), 7)
    assert path.endswith("/output/inference/run.epoch-0-flow-field/000007.flo")
    import os
    os.makedirs(os.path.dirname(path))
    open(path, "wb").write(io_golden["flo_bytes"].tobytes())
    prov = FloFlowProvider(str(img))
    assert prov.get_flow_uv(7).tobytes() == io_golden["flo_read"].tobytes()
    with pytest.raises(OSError):
        prov.get_flow_uv(8)
    open(flo_path(str(img), 9), "wb").write(b"\0" * 64)
    with pytest.raises(AssertionError):
        prov.get_flow_uv(9)


def test_frame_result_json_matches_the_reference_text(io_golden):
    """processor.py:83-84 writes json.dumps(utils.get_json(result), indent=4, sort_keys=True) per frame: same text, including the
    numpy integers and float32 values that get_json turns into strings and the NaN a 0/0 rate leaves behind."""
    import json
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
    text = json.dumps(utils.get_json(r), indent=4, sort_keys=True)
    assert text == str(io_golden["json_text"])
    assert json.dumps(utils.get_json(vars(r)), indent=4, sort_keys=True) == text      # the dict form gives the same document
