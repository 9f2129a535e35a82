"""CPU: the C-ABI library loads, exports every symbol include/apm.h declares, and
fails LOUDLY (no CPU fallback) when no HIP device is present."""
import ctypes
import os
import re

import pytest

import helpers as H


def _declared_functions():
    src = open(os.path.join(H.ROOT, "include", "apm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(apm_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    apm = H.pkg()
    assert os.path.exists(apm.LIB_PATH), "libapm_hip.so not built: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(apm.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "include/apm.h declares %s but libapm_hip.so does not export it" % name
    assert sorted(apm.ABI_SYMBOLS) == declared


def test_abi_version():
    assert H.pkg().load_library().apm_abi_version() == 1


def test_no_cpu_fallback_without_device():
    apm = H.pkg()
    if apm.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(apm.ApmError) as e:
        apm.ApmContext(1)
    assert e.value.status == -2  # APM_ERR_NO_DEVICE
    with pytest.raises(apm.ApmError):
        apm.ApmContext(device=0)


def test_product_never_links_the_oracle():
    """the shipped library and CLI must not reference oracle/ in any way."""
    for sub in ("csrc/apm_kernels.hip", "csrc/apm_runtime.hip", "csrc/apm_core.h", "csrc/apm_internal.h",
                "host/apm_parallel.c", "__init__.py", "workloads.py", "Makefile"):
        p = os.path.join(H.PKG_DIR, sub)
        if os.path.exists(p):
            assert "oracle" not in open(p).read().lower().replace("no cpu fallback", ""), p
