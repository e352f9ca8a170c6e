"""Root folder of an experiment and its data/ results/ logs/ children (reference: paths.py:5-23)."""
from __future__ import annotations

import os
from pathlib import Path

_root = Path(os.getcwd())


def setSageFolder(path) -> None:
    """Same name and meaning as paths.py:6-8: every later output goes under `path`."""
    global _root
    _root = Path(path)


def _child(name: str) -> Path:
    d = _root / name
    d.mkdir(parents=True, exist_ok=True)  # superset of the reference: also creates missing parents
    return d


def getDataFolder() -> Path:
    return _child("data")


def getResultsFolder() -> Path:
    return _child("results")


def getLogsFolder() -> Path:
    return _child("logs")
