"""Shared test helpers: package import, oracle binding, golden vectors.

The oracle (oracle/liboracle.so) is the CHECKER; it is only ever loaded here,
in __graft_entry__.smoke() and in bench.py's cpu_baseline leg.
"""
import base64
import ctypes
import importlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_NAME = "inf560-approximate-pattern-matching_amd"
PKG_DIR = os.path.join(ROOT, PKG_NAME)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_UTILS_SO = os.path.join(ROOT, "oracle", "_ref", "libref_utils.so")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "apm_sequential")


def pkg():
    return importlib.import_module(PKG_NAME)


def workloads():
    return importlib.import_module(PKG_NAME + ".workloads")


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True)
        lib = ctypes.CDLL(ORACLE_SO)
        c = ctypes
        lib.oracle_window_distance.restype = c.c_int
        lib.oracle_window_distance.argtypes = [c.c_char_p, c.c_char_p, c.c_int, c.POINTER(c.c_int)]
        for name in ("oracle_count_range_mt", "oracle_count_range_banded_mt"):
            fn = getattr(lib, name)
            fn.restype = c.c_int64
            fn.argtypes = [c.c_char_p, c.c_uint64, c.c_char_p, c.c_int, c.c_int, c.c_uint64, c.c_uint64, c.c_int]
        lib.oracle_count.restype = c.c_int64
        lib.oracle_count.argtypes = [c.c_char_p, c.c_uint64, c.c_char_p, c.c_int, c.c_int]
        lib.oracle_count_range.restype = c.c_int64
        lib.oracle_count_range.argtypes = [c.c_char_p, c.c_uint64, c.c_char_p, c.c_int, c.c_int, c.c_uint64, c.c_uint64]
        lib.oracle_max_threads.restype = c.c_int
        _oracle = lib
    return _oracle


def oracle_counts(text, patterns, k, banded=False, threads=0, j_begin=0, j_end=None):
    lib = oracle()
    fn = lib.oracle_count_range_banded_mt if banded else lib.oracle_count_range_mt
    n = len(text)
    if j_end is None:
        j_end = n
    out = []
    for p in patterns:
        r = fn(text, n, p, len(p), k, j_begin, j_end, threads)
        assert r >= 0
        out.append(r)
    return out


def window_distance(p, t):
    m = len(p)
    col = (ctypes.c_int * (m + 1))()
    return oracle().oracle_window_distance(p, t, m, col)


_golden = None


def golden():
    global _golden
    if _golden is None:
        with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
            g = json.load(f)
        for c in g["cases"]:
            c["patterns"] = [base64.b64decode(p) for p in c["patterns_b64"]]
            if "file" in c:
                c["path"] = os.path.join(GOLDEN_DIR, "dna", c["file"])
            else:
                c["text_bytes"] = base64.b64decode(c["text_b64"])
        _golden = g
    return _golden


def case_text(c):
    if "text_bytes" in c:
        return c["text_bytes"]
    with open(c["path"], "rb") as f:
        return f.read()


def case_cells(c):
    n = len(case_text(c)) if "text_bytes" in c else os.path.getsize(c["path"])
    return sum(max(0, n - c["k"]) * len(p) ** 2 for p in c["patterns"])
