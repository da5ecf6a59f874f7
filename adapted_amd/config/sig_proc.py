"""``SigProcConfig``: the nested signal-processing configuration and its loaders.

Same public names and behaviour as the reference (adapted/config/sig_proc.py:161-265,
adapted/config/base.py:113-174): exactly one primary method, derived ``sig_preload_size``,
unknown keys/sections raise ``ValueError``.  Unlike the reference loader, derived fields are
refreshed after loading (the reference leaves them stale until parser.py:253-254 re-calls
the ``update_*`` methods) and dataclass defaults use factories (Python >= 3.11 safe).
"""
from __future__ import annotations

import copy
import dataclasses
import logging
import os
import sys
from typing import Any, Dict, Optional

from . import toml_io
from .schema import (PRESETS, SECTION_CLASSES, SPEEDS, CNNBoundariesConfig, CoreConfig,
                     LLRBoundariesConfig, MedShiftConfig, MVSPolyAConfig, RealRangeConfig,
                     RNAStartPeakConfig, StreamingConfig, _Section)
from .._version import __version__


@dataclasses.dataclass
class SigProcConfig:
    core: Any = dataclasses.field(default_factory=CoreConfig)
    llr_boundaries: Any = dataclasses.field(default_factory=LLRBoundariesConfig)
    mvs_polya: Any = dataclasses.field(default_factory=MVSPolyAConfig)
    real_range: Any = dataclasses.field(default_factory=RealRangeConfig)
    streaming: Optional[Any] = None
    cnn_boundaries: Any = dataclasses.field(default_factory=CNNBoundariesConfig)
    med_shift: Any = dataclasses.field(default_factory=MedShiftConfig)
    rna_start_peak: Any = dataclasses.field(default_factory=RNAStartPeakConfig)
    primary_method: Optional[str] = None
    primary_config: Optional[Any] = None

    def __post_init__(self):
        self.update_primary_method()
        self.update_sig_preload_size()

    # -- derived fields -------------------------------------------------------------
    def update_sig_preload_size(self):
        extra = 0
        if self.mvs_polya.mvs_detect_check:
            extra = self.mvs_polya.search_window + max(self.mvs_polya.median_shift_window,
                                                       self.mvs_polya.polyA_window)
        self.sig_preload_size = self.core.max_obs_trace + extra

    def update_primary_method(self):
        flags = {
            "llr": bool(self.llr_boundaries.llr_detect),
            "cnn": bool(self.cnn_boundaries.cnn_detect),
            "start_peak": bool(self.rna_start_peak.detect_rna_start_peak),
        }
        if sum(flags.values()) != 1:
            raise ValueError("Exactly one primary method must be enabled")
        self.primary_method = next(k for k, v in flags.items() if v)
        self.primary_config = {"llr": self.llr_boundaries, "cnn": self.cnn_boundaries,
                               "start_peak": self.rna_start_peak}[self.primary_method]
        if self.primary_method == "cnn":
            self.check_cnn_downscale_factor()

    def check_cnn_downscale_factor(self):
        from ..detect.cnn import MODEL_DOWNSCALE

        want = MODEL_DOWNSCALE.get(os.path.basename(self.cnn_boundaries.model_name))
        if want is not None and want != self.core.downscale_factor:
            msg = "CNN downscale factor and core downscale factor do not match"
            logging.error(msg)
            raise ValueError(msg)

    # -- dict / TOML ------------------------------------------------------------------
    def dict(self):
        out = {}
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            out[f.name] = v.dict() if isinstance(v, _Section) else v
        out["primary_config"] = self.primary_config.dict() if self.primary_config is not None else None
        return out

    def typed_dict(self):
        out: Dict[str, Any] = {}
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            if f.name == "primary_config":
                v = self.primary_config
            out[f.name] = v.typed_dict() if isinstance(v, _Section) else v
        return out

    def to_toml(self, file_path: str):
        toml_io.dump(self.typed_dict(), file_path)

    def copy(self):
        return copy.deepcopy(self)

    def pretty_print(self, file=sys.stdout):
        import pprint

        for key, val in self.dict().items():
            if isinstance(val, dict):
                print("%s:\n%s" % (key, pprint.pformat(val, sort_dicts=False)), file=file)
            else:
                print("%s: %s" % (key, val), file=file)


def nested_config_from_dict(config_dict: Dict[str, Any], config_class=SigProcConfig) -> SigProcConfig:
    valid = [f.name for f in dataclasses.fields(config_class)]
    unknown = [k for k in config_dict if k not in valid]
    if unknown:
        msg = "Invalid config file. Unknown key(s): %s. Valid keys are: %s" % (", ".join(unknown), ", ".join(valid))
        logging.error(msg)
        raise ValueError(msg)
    sections = {}
    scalars = {}
    for key, content in config_dict.items():
        if isinstance(content, dict):
            if key == "primary_config":
                continue  # derived; written by to_toml for information only
            cls = SECTION_CLASSES.get(key)
            if cls is None:
                msg = "Invalid section type for %s" % key
                logging.error(msg)
                raise ValueError(msg)
            try:
                kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in content.items()}
                sections[key] = cls(**kw)
            except TypeError:
                msg = "Invalid config file. Could not parse section %s with content %s as %s" % (key, content, cls)
                logging.error(msg)
                raise ValueError(msg)
        else:
            scalars[key] = content
    # build without tripping the exactly-one check on half-populated defaults
    obj = config_class.__new__(config_class)
    for f in dataclasses.fields(config_class):
        if f.name in sections:
            setattr(obj, f.name, sections[f.name])
        elif f.default_factory is not dataclasses.MISSING:  # type: ignore[attr-defined]
            setattr(obj, f.name, f.default_factory())  # type: ignore[misc]
        else:
            setattr(obj, f.name, f.default)
    for k, v in scalars.items():
        if k not in ("primary_method",):
            setattr(obj, k, v)
    obj.update_primary_method()
    obj.update_sig_preload_size()
    return obj


def load_nested_config_from_file(file_path, config_class=SigProcConfig) -> SigProcConfig:
    return nested_config_from_dict(toml_io.load(file_path), config_class)


def chemistry_specific_config_name(chemistry: str, version: Optional[str] = None) -> str:
    version = version or __version__
    return "%s_%s@v%s" % (chemistry.lower(), SPEEDS[chemistry.lower()], version)


def config_name_to_dict(config_name: str) -> Dict[str, Any]:
    chem = config_name.split("_")[0].lower()
    return copy.deepcopy(PRESETS[chem])


def get_config(config_name: str) -> SigProcConfig:
    return nested_config_from_dict(config_name_to_dict(config_name))


def get_chemistry_specific_config(chemistry: str, version: Optional[str] = None) -> SigProcConfig:
    if chemistry.lower() not in PRESETS:
        msg = "Unknown chemistry: %s" % chemistry
        logging.error(msg)
        raise ValueError(msg)
    return get_config(chemistry_specific_config_name(chemistry, version))
