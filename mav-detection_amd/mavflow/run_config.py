"""RunConfig -- mode / dataset switches with the reference's enum surface (/root/reference/src/run_config.py:13-62).

Every member value is a 1-tuple, as the reference's trailing commas make them, and str(member) is the bare member name;
look-ups are by NAME (RunConfig.Mode[mode_key]).  The disk datasets (MIDGARD, AirSim, ...) are outside the hot path:
get_dataset() hands back a ready dataset object (e.g. mavflow.processor.SyntheticDataset) instead of opening files."""
from __future__ import annotations

import logging
from enum import Enum
from typing import Any, Dict, Type


class _Switch(Enum):
    """Enum whose text form is the member name alone (the reference strips the class prefix)."""

    def __str__(self) -> str:
        return self.name


def _switch(name: str, members: str) -> Type[_Switch]:
    return _Switch(name, {m: (i,) for i, m in enumerate(members.split())}, qualname=f"RunConfig.{name}", module=__name__)


def _by_name(kind: Type[_Switch], key: str, what: str) -> _Switch:
    if key in kind.__members__:
        return kind[key]
    raise ValueError(f"{what} {key} is not a valid {what.lower()} type, has to be one of {', '.join(kind.__members__)}")


class RunConfig:
    Mode = _switch("Mode", "APPEARANCE_RGB FLOW_UV FLOW_RADIAL FLOW_FOE_YOLO FLOW_FOE_CLUSTERING")
    DatasetType = _switch("DatasetType", "MIDGARD SIMULATION EXPERIMENT VIS_DRONE")

    _FIELDS = ("logger", "dataset", "sequence", "debug", "prepare_dataset", "validate", "headless", "data_to_yolo", "undistort")

    def __init__(self, logger: logging.Logger, dataset: Any, sequence: str, debug: bool, prepare_dataset: bool,
                 validate: bool, headless: bool, data_to_yolo: bool, undistort: bool, mode: str):
        given = (logger, dataset, sequence, debug, prepare_dataset, validate, headless, data_to_yolo, undistort)
        for field, value in zip(self._FIELDS, given):
            setattr(self, field, value)
        self.mode = self.get_mode(mode)
        self.results: Dict[int, Any] = {}
        self.settings: Dict[str, Any] = {}          # the reference reads settings.json here (sequence lists for the disk datasets)

    def get_mode(self, mode_key: str):
        return _by_name(RunConfig.Mode, mode_key, "Mode")

    def get_dataset_type(self, dataset_key: str):
        return _by_name(RunConfig.DatasetType, dataset_key.upper(), "Dataset")

    def uses_nn_for_detection(self) -> bool:
        M = RunConfig.Mode
        return self.mode in (M.FLOW_UV, M.FLOW_RADIAL, M.FLOW_FOE_YOLO)

    def get_dataset(self):
        if isinstance(self.dataset, str):
            raise NotImplementedError("disk datasets (ffmpeg / Docker FlowNet2 pipelines) are outside the hot path; pass a "
                                      "dataset object such as mavflow.processor.SyntheticDataset")
        return self.dataset

    def __str__(self) -> str:
        return f"{self.dataset}/{self.sequence}/{self.mode}"
