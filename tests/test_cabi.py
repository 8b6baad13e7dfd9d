"""CPU tests of the boundary: the C-ABI library builds for gfx950, loads without a GPU and
exports every entry point include/dvda_mlp_hip.h declares; no compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "dvda_mlp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dvda_(?:mlp_|pcm_)?hip_\w+)\s*\(", text)))


def test_header_declares_the_batch_tier():
    names = declared_functions()
    for must in ("dvda_mlp_hip_create", "dvda_mlp_hip_destroy", "dvda_mlp_hip_index",
                 "dvda_mlp_hip_decode", "dvda_mlp_hip_stream_info"):
        assert must in names


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.hipdec.lib()
    for name in declared_functions():
        assert hasattr(lib, name), "missing export: " + name
    assert b"gfx950" in lib.dvda_mlp_hip_version()


def test_binding_lists_the_same_exports(pkg):
    assert set(pkg.hipdec.EXPORTS) <= set(declared_functions())


def test_code_object_targets_gfx950(pkg):
    so = pkg._build.HIP_SO
    data = open(so, "rb").read()
    assert b"gfx950" in data
    assert b"k_decode" in data


def test_no_cpu_fallback_without_gpu(pkg):
    """On a machine without a GPU the product path must fail loudly, not decode on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = pkg.synth.make_cfg(n_aus=8)
    data, _ = pkg.synth.stream(cfg, 1)
    with pytest.raises(pkg.hipdec.HipError):
        pkg.hipdec.decode_streams([data])
    h = ctypes.c_void_p()
    assert pkg.hipdec.lib().dvda_mlp_hip_create(ctypes.byref(h), 0, 1, 16) != 0


def test_product_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkgdir = os.path.join(ROOT, "libdvd-audio_amd")
    for base, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle_lib" not in text and "libmlp_oracle" not in text and "libdvda_ref" not in text, f


def test_inline_asm_selects_keep_the_sgpr_hazard_distance():
    """The decode kernels hold a few v_cndmask selects as inline asm so that their compares can be
    issued ahead of them; LLVM does not pad inline asm for the gfx940+ 'VALU writes SGPR -> VALU
    reads SGPR' hazard (2 wait states, no hardware interlock).  tools/hazard_check.py compiles the
    kernels to assembly and verifies the distance for every such select."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hazard_check.py")], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "hazard violations: 0" in r.stdout


def test_c_shard_is_the_python_shard():
    """dvda_mlp_hip_shard (csrc/mlp_multi.cpp, what a C host deals its title list to the GPUs with) computes the
    partition libdvd-audio_amd/shard.py computes -- greedy longest-processing-time, ties included."""
    import numpy as np
    import libdvd_audio_amd as pkg
    rng = np.random.default_rng(5)
    for n, parts in ((1, 1), (7, 2), (64, 8), (1000, 3), (33, 5)):
        sizes = rng.integers(1, 50, n).astype(np.int64) * 1000        # (many ties)
        owner = pkg.hipdec.shard_c(sizes, parts)
        for r in range(parts):
            assert np.array_equal(np.flatnonzero(owner == r), pkg.shard.shard_titles(sizes, parts, r))
