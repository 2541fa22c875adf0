"""Host-side helpers with the reference's names and semantics (/root/reference/src/utils.py):
Rectangle (:13-104), line_intersection (:183-197), .flo read/write (:204-257), get_json (:350-361).
These are plain Python/numpy glue (a handful of scalars per frame), not part of the GPU path."""
from __future__ import annotations

import json
from typing import Any, Dict, List, Tuple

import numpy as np

Point = Tuple[float, float]


class Rectangle:
    """topleft + size box.  from_points keeps the reference's inclusive-end convention: size = bottomright - topleft."""

    def __init__(self, topleft: Point, size: Point) -> None:
        self.topleft: Point = topleft
        self.size: Point = size

    @classmethod
    def from_center(cls, center: Point, size: Point) -> "Rectangle":
        return cls((center[0] - size[0] / 2, center[1] - size[1] / 2), size)

    @classmethod
    def from_points(cls, topleft: Point, bottomright: Point) -> "Rectangle":
        return cls(topleft, (bottomright[0] - topleft[0], bottomright[1] - topleft[1]))

    @classmethod
    def from_yolo_input(cls, arr: List[float], img_size: np.ndarray) -> "Rectangle":
        dims = img_size.astype(np.float64)
        return cls.from_center(np.array([arr[1], arr[2]]) * dims, np.array([arr[3], arr[4]]) * dims)

    @classmethod
    def from_yolo_output(cls, arr: List[float]) -> "Rectangle":
        return cls((arr[0], arr[1]), (arr[2], arr[3]))

    @classmethod
    def from_box(cls, box) -> "Rectangle":
        """From libmavflow's (x0, y0, x1, y1) inclusive record; (-1,-1,-1,-1) gives topleft (-1,-1), size (0,0)."""
        x0, y0, x1, y1 = (int(v) for v in box)
        return cls.from_points((x0, y0), (x1, y1))

    def get_topleft(self) -> Point:
        return (self.topleft[0], self.topleft[1])

    def get_bottomright(self) -> Point:
        return (self.topleft[0] + self.size[0], self.topleft[1] + self.size[1])

    def get_topleft_int(self) -> Tuple[int, int]:
        return (int(self.topleft[0]), int(self.topleft[1]))

    def get_topleft_int_offset(self) -> Tuple[int, int]:
        return (int(self.topleft[0]), int(self.topleft[1]) - 5)

    def get_bottomright_int(self) -> Tuple[int, int]:
        return (int(self.topleft[0] + self.size[0]), int(self.topleft[1] + self.size[1]))

    def get_center(self) -> Point:
        return (self.topleft[0] + self.size[0] / 2, self.topleft[1] + self.size[1] / 2)

    def get_center_int(self) -> Tuple[int, int]:
        cx, cy = self.get_center()
        return (int(cx), int(cy))

    def get_left(self) -> float:
        return self.topleft[0]

    def get_right(self) -> float:
        return self.topleft[0] + self.size[0]

    def get_top(self) -> float:
        return self.topleft[1]

    def get_bottom(self) -> float:
        return self.topleft[1] + self.size[1]

    def get_area(self) -> float:
        return max(1.0, self.size[0] * self.size[1])     # floored at 1.0 like the reference (:78-79)

    def to_yolo(self, img_size: np.ndarray, obj_id: int = 0) -> str:
        dims = img_size.astype(np.float64)
        c = np.array(self.get_center()) / dims
        s = np.array(self.size) / dims
        return f"{obj_id} {c[0]} {c[1]} {s[0]} {s[1]}\n"

    @classmethod
    def calculate_iou(cls, r1: "Rectangle", r2: "Rectangle") -> float:
        w = min(r1.get_right(), r2.get_right()) - max(r1.get_left(), r2.get_left())
        h = min(r1.get_bottom(), r2.get_bottom()) - max(r1.get_top(), r2.get_top())
        overlap = w * h
        return overlap / (r1.get_area() + r2.get_area() - overlap)


def line_intersection(line1, line2):
    """Intersection of the lines through line1 = (p, q) and line2 = (p, q); (False, False) when they are parallel."""
    (ax, ay), (bx, by) = line1
    (cx, cy), (dx, dy) = line2
    xd = (ax - bx, cx - dx)
    yd = (ay - by, cy - dy)
    div = xd[0] * yd[1] - xd[1] * yd[0]
    if div == 0:
        return False, False
    d0 = ax * by - ay * bx
    d1 = cx * dy - cy * dx
    return (d0 * xd[1] - d1 * xd[0]) / div, (d0 * yd[1] - d1 * yd[0]) / div


FLO_TAG = 202021.25


def read_flow(filename: str) -> np.ndarray:
    """Middlebury .flo -> float32 (h, w, 2).  Bad tag -> AssertionError, as the reference (:217)."""
    with open(filename, "rb") as f:
        tag = np.fromfile(f, np.float32, count=1)
        assert tag.size == 1 and tag[0] == np.float32(FLO_TAG), "Flow number %r incorrect. Invalid .flo file" % (tag,)
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        data = np.fromfile(f, np.float32, count=2 * w * h)
    return np.resize(data, (h, w, 2))


def write_flow(filename: str, uv: np.ndarray, v: np.ndarray = None) -> None:
    """float (h, w, 2) (or separate u, v planes) -> Middlebury .flo."""
    if v is None:
        assert uv.ndim == 3 and uv.shape[2] == 2
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u = uv
    assert u.shape == v.shape
    h, w = u.shape
    with open(filename, "wb") as f:
        np.array([FLO_TAG], np.float32).tofile(f)
        np.array([w, h], np.int32).tofile(f)
        np.stack([u, v], axis=-1).astype(np.float32).tofile(f)


def get_json(obj: Dict[str, Any]) -> Dict[str, Any]:
    """JSON-safe dictionary; anything json cannot encode becomes its __dict__ or its str() (numpy ints -> strings)."""
    return json.loads(json.dumps(obj, default=lambda o: getattr(o, "__dict__", str(o))))


def assert_type(x):
    assert x is not None
    return x
