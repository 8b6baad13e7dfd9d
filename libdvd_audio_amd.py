"""Import alias: `import libdvd_audio_amd` -> the package in ./libdvd-audio_amd/."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("libdvd-audio_amd")
sys.modules[__name__] = _pkg
