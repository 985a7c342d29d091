"""CPU: the C-ABI library builds, loads and exports every symbol include/goofer_hip.h declares."""
import ctypes
import os
import re

import numpy as np

from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, "include", "goofer_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(goofer_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from goofer_amd import build
    path = build.build()
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n


def test_binding_covers_header_and_struct_layout():
    from goofer_amd import _lib
    assert set(_declared()) <= set(_lib.EXPORTS)
    lib = _lib.load()
    assert lib.goofer_version().startswith(b"goofer_hip")
    assert _lib.NOTE_PARAMS.itemsize == lib.goofer_sizeof(0) == 112
    assert ctypes.sizeof(_lib.Batch) == lib.goofer_sizeof(1)
    assert _lib.NOTE_PLAN.itemsize == lib.goofer_sizeof(2)
    assert ctypes.sizeof(_lib.Assembly) == lib.goofer_sizeof(3)
    assert _lib.ONEPOLE_JOB.itemsize == lib.goofer_sizeof(4)
    assert _lib.POST_NOTE.itemsize == lib.goofer_sizeof(5)
    assert ctypes.sizeof(_lib.Post) == lib.goofer_sizeof(6)
    assert _lib.PLAN_REQUEST.itemsize == lib.goofer_sizeof(7)
    assert _lib.PLAN_GEOMETRY.itemsize == lib.goofer_sizeof(8)


def test_no_cpu_fallback_without_gpu():
    import pytest
    import torch
    from goofer_amd.device import Context, GooferError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(GooferError):
        Context(0)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import pytest
    from goofer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load(str(tmp_path / "nope.so"))


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "goofer_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "from . import oracle" not in src and "/root/reference" not in src, f


def test_c_host_example_compiles_and_links(tmp_path):
    """examples/c_host.c is a torch-free C99 host for the ABI: it must compile with plain gcc against include/ and link
    against the built library (it runs only where an MI355X is visible)."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "goofer_amd", "libgoofer_hip.so")
    if not os.path.exists(lib) or shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs the built library, gcc and the ROCm headers")
    out = tmp_path / "c_host"
    cmd = ["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "examples", "c_host.c"),
           "-I" + os.path.join(root, "include"), "-I/opt/rocm/include", "-L" + os.path.join(root, "goofer_amd"), "-lgoofer_hip",
           "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + os.path.join(root, "goofer_amd"), "-Wl,-rpath,/opt/rocm/lib",
           "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.exists()


def test_core_exposes_the_reference_module_surface():
    """Every GOOFER.py name that SillySampler.py / SillyEditor.py / test.py reach through `gf.` (SURVEY §8 b)."""
    from goofer_amd import core
    for name in ("extract_features", "synthesize", "save_features", "load_features", "decode_env_from_knots", "interp1d",
                 "gaussian_filter1d", "gaussian_filter", "stretch_feature", "create_volume_jitter", "rms", "pulse_train_numba",
                 "stft", "istft", "compress_env_to_knots", "make_mel_knots", "formants_to_int_keys", "to_compute"):
        assert callable(getattr(core, name)), name


def test_stft_refuses_a_window_that_is_not_the_reference_window():
    """gf.stft / gf.istft take a ``window`` (GOOFER.py:355, 392); the kernels use the plan's sqrt-Hann.  The reference's own window
    passes, any other raises before anything touches the device (it used to be ignored silently: VERDICT r5)."""
    import numpy as np
    import pytest
    from goofer_amd import core
    n_fft = 1024
    core._check_window(None, n_fft)
    core._check_window(np.sqrt(np.hanning(n_fft)).astype(np.float32), n_fft)
    core._check_window(np.sqrt(np.hanning(n_fft)), n_fft)                       # the same window in fp64
    for bad in (np.hanning(n_fft), np.ones(n_fft, np.float32), np.sqrt(np.hanning(n_fft // 2)), np.sqrt(np.hamming(n_fft))):
        with pytest.raises(ValueError):
            core._check_window(bad, n_fft)
    with pytest.raises(ValueError):
        core.stft(np.zeros(4096, np.float32), n_fft=n_fft, hop_length=256, window=np.ones(n_fft, np.float32))
    with pytest.raises(ValueError):
        core.istft(np.zeros((513, 8), np.complex64), hop_length=256, window=np.ones(n_fft, np.float32))
