#!/usr/bin/env python3
"""Builds surge_amd/libsurge_replay_exp.so: the library with -DSURGE_EXPERIMENTS (experiment hooks that can change results,
e.g. SURGE_REPLAY_RTC_EXTRA; never compiled into libsurge_replay.so).  Load it with SURGE_REPLAY_LIB=<path>."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surge_amd import _native

_native.FLAGS = _native.FLAGS + ("-DSURGE_EXPERIMENTS",)
_native.OBJ_DIR = os.path.join(os.path.dirname(_native.OBJ_DIR), "build_exp")
_native.LIB_PATH = os.path.join(os.path.dirname(_native.LIB_PATH), "libsurge_replay_exp.so")
print(_native.build(force=True))
