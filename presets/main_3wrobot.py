#!/usr/bin/env python3
"""Preset: 3wrobot on the MI355X-native path.  Same command-line flags and defaults as the reference's
presets/main_3wrobot.py; the loop is the reference's headless loop (always headless here).  Example:

    python presets/main_3wrobot.py --ctrl_mode MPC --t1 2.0 --batch 64
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from rcognita_amd.presets import run  # noqa: E402

if __name__ == "__main__":
    run("3wrobot")
