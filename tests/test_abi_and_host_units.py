"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/cblx.h declares (no
compute calls without a GPU), fails loudly without a device, and the scalar device/host code shared with the
kernels (cbl_amd/csrc/necklace.hpp) agrees with the definition."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

import cbl_amd

ROOT = Path(__file__).resolve().parent.parent


def _header_functions():
    text = (ROOT / "include" / "cblx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cblx_[a-z0-9_]+)\s*\(", text)) - {"cblx_bucket_cb"})


def test_library_exports_every_declared_symbol():
    L = cbl_amd.lib()
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/cblx.h but not exported by libcblx.so"
    assert sorted(cbl_amd.SIGNATURES) == names
    assert L.cblx_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cbl_amd.CblxError) as e:
        cbl_amd.CBL(31, 24)
    assert e.value.code == cbl_amd.EDEVICE


def test_parameter_validation_needs_no_gpu():
    for k, pb in ((30, 24), (3, 2), (61, 24), (31, 0), (31, 33), (7, 18)):
        with pytest.raises(cbl_amd.CblxError) as e:
            cbl_amd.CBL(k, pb)
        assert e.value.code == cbl_amd.EINVAL, (k, pb)


def test_necklace_host_unit(tmp_path):
    exe = tmp_path / "necklace_unit"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), str(ROOT / "tests" / "host" / "necklace_unit.cpp")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
