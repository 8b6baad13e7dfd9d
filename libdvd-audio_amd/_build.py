"""In-tree build of the native pieces (hipcc / gcc, no network, no JIT cache).

    libdvda_mlp_hip.so   HIP kernels + C ABI (include/dvda_mlp_hip.h), gfx950
    libdvd_audio_hip.so  disc-level API (include/dvd-audio-hip.h), plain C on top of the above
    libmlp_synth.so      synthetic MLP stream generator (tooling for tests/bench)

The oracle (oracle/Makefile) is built by __graft_entry__.build(), not here: it is
test infrastructure and the product never loads it.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(HERE, "libdvda_mlp_hip.so")
SYNTH_SO = os.path.join(HERE, "synth", "libmlp_synth.so")
DISC_SO = os.path.join(HERE, "libdvd_audio_hip.so")
DISC_SRCS = [os.path.join(HERE, "csrc", "dvda_disc.c"),
             os.path.join(os.path.dirname(HERE), "include", "dvd-audio-hip.h"),
             os.path.join(os.path.dirname(HERE), "include", "dvda_mlp_hip.h")]

# [0] the HIP translation unit, [1] the streaming tier; behind them EVERY header under csrc/ (the staleness
# test looks at all of them: an edit to any header mlp_hip.hip includes rebuilds the library)
HIP_SRCS = [os.path.join(HERE, "csrc", "mlp_hip.hip"), os.path.join(HERE, "csrc", "mlp_stream.c"),
            os.path.join(HERE, "csrc", "mlp_multi.cpp")]
HIP_SRCS += sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".h"))
HIP_SRCS.append(os.path.join(os.path.dirname(HERE), "include", "dvda_mlp_hip.h"))
SYNTH_SRCS = [os.path.join(HERE, "synth", f) for f in ("mlp_synth.c", "mlp_synth.h")]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X decode path cannot be built")


def build_hip(force=False, verbose=False, defines=(), out=None):
    """defines/out: diagnostic variants for tools/ab_bench.sh (never the shipped library)."""
    target = out or HIP_SO
    if not force and not _stale(target, HIP_SRCS):
        return target
    # the kernels + batch tier are HIP C++; the streaming tier (mlp.h mirror) is plain C
    tag = os.path.basename(target).replace(".so", "")
    obj_c = os.path.join(HERE, "csrc", "mlp_stream_%s.o" % tag)
    obj_hip = os.path.join(HERE, "csrc", "mlp_hip_%s.o" % tag)
    subprocess.run(["gcc", "-O2", "-fPIC", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-c", "-o", obj_c, HIP_SRCS[1]], check=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c",
           "-o", obj_hip, HIP_SRCS[0]] + ["-D" + d for d in defines if not d.startswith("-")] + \
          [d for d in defines if d.startswith("-")]          # diagnostic builds may pass raw flags
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    # the multi-device dispatcher: host-only C++ on top of the entry points above
    obj_multi = os.path.join(HERE, "csrc", "mlp_multi_%s.o" % tag)
    subprocess.run(["g++", "-O2", "-fPIC", "-Wall", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-c", "-o", obj_multi, HIP_SRCS[2]], check=True)
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", target, obj_hip, obj_c, obj_multi,
                    "-lpthread"], check=True)
    return target


BOUNDS_SO = os.path.join(HERE, "libdvda_mlp_hip_bounds.so")


def build_bounds(force=False):
    """The range-checked diagnostic build (csrc/mlp_bounds.h; tests/test_gpu_soak.py loads it through
    DVDA_MLP_HIP_LIB).  Never the shipped library."""
    if not force and not _stale(BOUNDS_SO, HIP_SRCS):
        return BOUNDS_SO
    return build_hip(force=True, defines=["DVDA_BOUNDS=1"], out=BOUNDS_SO)


def build_disc(force=False):
    """tier C (dvd-audio.h mirror): links against libdvda_mlp_hip.so next to it."""
    if not force and not _stale(DISC_SO, DISC_SRCS + [HIP_SO]):
        return DISC_SO
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-o", DISC_SO, DISC_SRCS[0], "-L" + HERE, "-ldvda_mlp_hip", "-L/opt/rocm/lib", "-lamdhip64",
                    "-lm", "-lpthread", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return DISC_SO


TOOL = os.path.join(os.path.dirname(HERE), "build", "dvda2wav_hip")
TOOL_SRC = os.path.join(os.path.dirname(HERE), "tools", "dvda2wav_hip.c")


def build_tool(force=False):
    """dvda2wav_hip: the command-line extractor on top of libdvd_audio_hip.so."""
    if not force and not _stale(TOOL, [TOOL_SRC, DISC_SO, DISC_SRCS[1]]):
        return TOOL
    os.makedirs(os.path.dirname(TOOL), exist_ok=True)
    subprocess.run(["gcc", "-O2", "-Wall", "-o", TOOL, TOOL_SRC, "-I" + os.path.join(os.path.dirname(HERE), "include"),
                    "-L" + HERE, "-ldvd_audio_hip", "-ldvda_mlp_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
                    "-Wl,-rpath,$ORIGIN/../libdvd-audio_amd", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return TOOL


def build_synth(force=False):
    if not force and not _stale(SYNTH_SO, SYNTH_SRCS):
        return SYNTH_SO
    subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-o", SYNTH_SO, SYNTH_SRCS[0],
                    "-lpthread"], check=True)
    return SYNTH_SO


def build_all(force=False, verbose=False):
    # (the range-checked diagnostic library is built by its own test, tests/test_gpu_soak.py -> build_bounds(): a second
    #  compile of the heaviest translation unit is not every user's business)
    return build_hip(force, verbose), build_disc(force), build_tool(force), build_synth(force)
