"""libdvd-audio_amd -- MI355X-native MLP (Meridian Lossless Packing) decode path.

Only the hot path of tuffy/libdvd-audio lives here:

    csrc/      hand-written gfx950 HIP kernels + the C ABI of include/dvda_mlp_hip.h
    hipdec.py  ctypes binding of that C ABI (device memory via torch)
    discdec.py ctypes binding of the disc-level API (include/dvd-audio-hip.h, csrc/dvda_disc.c)
    synth/     synthetic MLP stream generator (tooling for tests and bench)

The directory name carries a hyphen (it mirrors the reference's name); import it as
`import libdvd_audio_amd` (the alias module at the repo root) or with importlib.
"""
from . import _build  # noqa: F401
from . import hipdec, shard, synth  # noqa: F401
from . import disc, discdec  # noqa: F401

__all__ = ["hipdec", "discdec", "shard", "synth", "_build"]
