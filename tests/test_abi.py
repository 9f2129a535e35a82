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


REFSHIM_SYMBOLS = ["getDeviceCount", "setDevice", "invoke_kernel", "write_kernel_result", "initializeGPU", "getGPUResult"]


def test_library_exports_the_references_own_gpu_entry_points():
    """include/apm_refshim.h: the six extern "C" symbols the reference's host files call
    (src/main.c:18-19, src/patterns_over_ranks.c:33-36, src/database_over_ranks.c:18-22) -> link unmodified."""
    lib = ctypes.CDLL(H.pkg().LIB_PATH)
    hdr = open(os.path.join(H.ROOT, "include", "apm_refshim.h")).read()
    for name in REFSHIM_SYMBOLS:
        assert hasattr(lib, name), name
        assert re.search(r"\b%s\s*\(" % name, hdr), name


def test_product_library_has_no_measurement_switches():
    """Stage-skipping and sizing knobs exist only under -DAPM_MEASURE (make measure -> libapm_hip_measure.so):
    the shipped library must not even contain their names, so no environment variable can change its results."""
    blob = open(H.pkg().LIB_PATH, "rb").read()
    for name in (b"APM_FILTER_ABLATE", b"APM_MEASURE_SKIP", b"APM_MAX_KEYS", b"APM_BPC_CAP", b"APM_QCAP_S1", b"APM_FUSED_THREADS", b"APM_VERIFY_GRID_PCT"):
        assert name not in blob, name
    for sub in ("csrc/apm_kernels.hip", "csrc/apm_runtime.hip", "csrc/apm_sieve.hip"):
        p = os.path.join(H.PKG_DIR, sub)
        if not os.path.exists(p):
            continue
        depth, measure_depth = 0, None
        for line in open(p):
            t = line.strip()
            if t.startswith("#if"):
                depth += 1
                if "APM_MEASURE" in t and measure_depth is None:
                    measure_depth = depth
            elif t.startswith("#endif"):
                if measure_depth == depth:
                    measure_depth = None
                depth -= 1
            elif measure_depth is None:
                assert "getenv(\"APM_MEASURE" not in line and "skip_mask" not in line, (sub, line)


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
