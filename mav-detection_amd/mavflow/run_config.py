"""RunConfig -- mode / dataset switches with the reference's enum surface (/root/reference/src/run_config.py:13-62).

Every member value is a 1-tuple, exactly as the reference's trailing commas make them; look-ups are by NAME
(RunConfig.Mode[mode_key]).  The disk datasets (MIDGARD, AirSim, ...) are outside the hot path: get_dataset()
accepts a ready dataset object (e.g. mavflow.processor.SyntheticDataset) instead of opening files."""
from __future__ import annotations

import logging
from enum import Enum
from typing import Any, Dict


class RunConfig:
    class Mode(Enum):
        APPEARANCE_RGB = 0,
        FLOW_UV = 1,
        FLOW_RADIAL = 2,
        FLOW_FOE_YOLO = 3,
        FLOW_FOE_CLUSTERING = 4,

        def __str__(self) -> str:
            return super().__str__().replace("Mode.", "")

    class DatasetType(Enum):
        MIDGARD = 0,
        SIMULATION = 1,
        EXPERIMENT = 2,
        VIS_DRONE = 3,

        def __str__(self) -> str:
            return super().__str__().replace("DatasetType.", "")

    def __init__(self, logger: logging.Logger, dataset: Any, sequence: str, debug: bool, prepare_dataset: bool,
                 validate: bool, headless: bool, data_to_yolo: bool, undistort: bool, mode: str):
        self.logger = logger
        self.dataset = dataset
        self.sequence = sequence
        self.debug = debug
        self.prepare_dataset = prepare_dataset
        self.validate = validate
        self.headless = headless
        self.data_to_yolo = data_to_yolo
        self.undistort = undistort
        self.mode = self.get_mode(mode)
        self.results: Dict[int, Any] = dict()
        self.settings: Dict[str, Any] = {}

    def get_mode(self, mode_key: str) -> "RunConfig.Mode":
        names = [m.name for m in RunConfig.Mode]
        if mode_key not in names:
            raise ValueError(f"Mode {mode_key} is not a valid mode type, has to be one of {', '.join(names)}")
        return RunConfig.Mode[mode_key]

    def get_dataset_type(self, dataset_key: str) -> "RunConfig.DatasetType":
        names = [m.name for m in RunConfig.DatasetType]
        key = dataset_key.upper()
        if key not in names:
            raise ValueError(f"Dataset {key} is not a valid dataset type, has to be one of {', '.join(names)}")
        return RunConfig.DatasetType[key]

    def uses_nn_for_detection(self) -> bool:
        return self.mode in (RunConfig.Mode.FLOW_UV, RunConfig.Mode.FLOW_RADIAL, RunConfig.Mode.FLOW_FOE_YOLO)

    def __str__(self) -> str:
        return f"{self.dataset}/{self.sequence}/{self.mode}"

    def get_dataset(self):
        if isinstance(self.dataset, str):
            raise NotImplementedError("disk datasets (ffmpeg / Docker FlowNet2 pipelines) are outside the hot path; pass a "
                                      "dataset object such as mavflow.processor.SyntheticDataset")
        return self.dataset
