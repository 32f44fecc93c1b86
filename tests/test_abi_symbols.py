"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
that include/cartpolepp_abi.h declares, the ctypes table covers exactly that set, and the library fails
loudly (no CPU fallback) when asked for a device that is not there."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cartpolepp_abi.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cpp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_families():
    syms = declared_symbols()
    for family in ("cpp_ctx_", "cpp_net_", "cpp_batch_", "cpp_replay_", "cpp_ddpg_", "cpp_prof_"):
        assert any(s.startswith(family) for s in syms), family
    assert len(syms) >= 45


def test_library_exports_every_declared_symbol_and_binding_matches():
    from cartpoleplusplus_amd import _lib
    syms = declared_symbols()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), "libcartpolepp_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.lib.cpp_abi_version() == 1


def test_every_entry_point_cites_the_reference():
    text = open(HEADER).read()
    for cite in ("replay_memory.py:", "base_network.py:", "ddpg_cartpole.py:", "util.py:"):
        assert cite in text


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cartpoleplusplus_amd import _lib
    with pytest.raises(RuntimeError) as e:
        _lib.Context(0)
    assert "hipGetDeviceCount" in str(e.value) or "device" in str(e.value)
    with pytest.raises(RuntimeError):
        from cartpoleplusplus_amd.replay_memory import ReplayMemory
        _lib.set_default_context(None)
        ReplayMemory(8, (2, 3), 2)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cartpoleplusplus_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_release_library_reads_no_environment_variable():
    """the CPP_* kernel-selection switches exist only in the ablation build (csrc/common.h: cpp_switch_off): the release
    library must not even import getenv; the ablation library, built from the same sources, does."""
    import subprocess
    from cartpoleplusplus_amd import _lib
    nm = "/opt/rocm/lib/llvm/bin/llvm-nm" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-nm") else "nm"
    rel = subprocess.check_output([nm, "-D", "--undefined-only", _lib.LIB_PATH]).decode()
    assert "getenv" not in rel
    abl = os.path.join(os.path.dirname(_lib.LIB_PATH), "libcartpolepp_hip_ablation.so")
    assert "getenv" in subprocess.check_output([nm, "-D", "--undefined-only", abl]).decode()
    for src in os.listdir(os.path.join(ROOT, "cartpoleplusplus_amd", "csrc")):
        if src.endswith((".hip", ".cpp", ".h")) and src != "common.h":
            assert "getenv" not in open(os.path.join(ROOT, "cartpoleplusplus_amd", "csrc", src)).read(), src
