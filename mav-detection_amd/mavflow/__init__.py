"""mavflow -- host-side mirror of the reference's per-frame-pair interface over libmavflow.so (HIP, gfx950).

    from mavflow import Farneback, FocusOfExpansion, Detector, Processor, RunConfig, FrameResult

Same class / method names and argument meaning as evroon/mav-detection's src/*.py for the hot path; the arithmetic
runs in hand-written HIP kernels behind the C-ABI of include/mavflow.h.  Nothing here imports oracle/."""


def __getattr__(name):
    import importlib
    table = {"Farneback": "farneback", "FocusOfExpansion": "focus_of_expansion", "Detector": "detector",
             "Processor": "processor", "SyntheticDataset": "processor", "RunConfig": "run_config",
             "FrameResult": "frame_result", "Rectangle": "utils", "Context": "_lib"}
    if name in table:
        return getattr(importlib.import_module(f"{__name__}.{table[name]}"), name)
    raise AttributeError(name)
