#!/usr/bin/env python3
"""Builds diagnostic variants of the HIP library: tools/ab_build.py name=DEFINE[,DEFINE] ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libdvd_audio_amd as pkg  # noqa: E402

for spec in sys.argv[1:]:
    name, _, defs = spec.partition("=")
    out = os.path.join(ROOT, "libdvd-audio_amd", "exp_%s.so" % name)
    pkg._build.build_hip(force=True, defines=[d for d in defs.split(",") if d], out=out)
    print("built", out)
