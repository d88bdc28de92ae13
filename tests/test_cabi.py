"""The C-ABI library loads on a machine without a GPU, exports every symbol include/safemotions_hip.h
declares, agrees with the ctypes mirror of smr_scene_t, and refuses to run without a device."""
import ctypes as C
import os
import re

import pytest

from safemotionsrisk_amd import lib
from safemotionsrisk_amd.scene import Scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "safemotions_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smr_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    L = lib.load()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n
    assert sorted(lib.EXPORTS) == names


def test_abi_version_and_struct_size(built):
    L = lib.load()
    assert L.smr_abi_version() == lib.ABI_VERSION
    assert L.smr_scene_sizeof() == C.sizeof(Scene)
    from oracle import oracle
    assert oracle.lib().smo_scene_sizeof() == C.sizeof(Scene)


def test_create_without_gpu_fails_loudly(built, space_scene):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = lib.load()
    h = C.c_void_p()
    rc = L.smr_create(C.cast(space_scene.pointer(), C.c_void_p), 8, 0, 0, 0, None, C.byref(h))
    assert rc == -3 and not h.value
    assert b"no HIP device" in L.smr_last_error(None)
    from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
    with pytest.raises(RuntimeError):
        VecSafeMotionsEnv(8, **SPACE_ENV_CONFIG)


def test_create_rejects_bad_scenes(built, space_scene):
    L = lib.load()
    h = C.c_void_p()
    assert L.smr_create(None, 8, 0, 0, 0, None, C.byref(h)) == -1
    assert L.smr_create(C.cast(space_scene.pointer(), C.c_void_p), 0, 0, 0, 0, None, C.byref(h)) == -1
    assert L.smr_create(C.cast(space_scene.pointer(), C.c_void_p), 8, 0xFFFFFFFC, 0, 0, None, C.byref(h)) == -1     # base + B beyond 32 bits
    assert b"env_index_base" in L.smr_last_error(None)
    space_scene.struct.version += 1
    try:
        assert L.smr_create(C.cast(space_scene.pointer(), C.c_void_p), 8, 0, 0, 0, None, C.byref(h)) == -1
        assert b"version" in L.smr_last_error(None)
    finally:
        space_scene.struct.version -= 1


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "safemotionsrisk_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "smr_oracle" not in text, f


def test_library_is_bound_after_pytorch():
    """PyTorch ships its own HIP runtime: lib.load() has to pull torch in first, or libsmr_hip.so binds the system copy
    and the process ends up with two runtimes that do not share devices, streams or memory (seen as 'no HIP device'
    and as kernels racing torch's copies when this test file alone ran on the GPU box)."""
    import subprocess, sys
    code = ("import sys; from safemotionsrisk_amd import lib; assert 'torch' not in sys.modules; lib.load(); "
            "assert 'torch' in sys.modules; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-500:]
