"""MI355X-native YOLOv3/v4 detection engine.

The directory doubles as the import root of the reference-compatible surface: put it on ``sys.path``
(``activate()`` does that) and ``import models``, ``from utils.utils import *``, ``import engine``
behave like the reference's top-level modules, with the hot path executed by ``libyolo_hip.so``.
"""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))


def activate():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return ROOT
