"""CPU: the oracle restatement against the reference's own golden vectors."""
import ctypes
import os
import random

import pytest

import helpers as H

CASES = H.golden()["cases"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_literal_matches_reference(case):
    """oracle_count (literal restatement of utils.c:76-99 + sequential.c:105-144)
    == counts printed by the reference binary."""
    if H.case_cells(case) > 6e10:
        pytest.skip("literal DP too slow for the CPU suite; covered by the banded variant")
    got = H.oracle_counts(H.case_text(case), case["patterns"], case["k"])
    assert got == case["counts"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_banded_matches_reference(case):
    """banded (|x-y| <= k/2) predicate == reference counts on EVERY golden case."""
    got = H.oracle_counts(H.case_text(case), case["patterns"], case["k"], banded=True)
    assert got == case["counts"]


def test_readme_published_counts():
    """README.md:58-63 of the reference: config #1 prints 0,4,4,4,4,4."""
    c = next(c for c in CASES if c["name"] == "cfg1_basic_test")
    assert c["counts"] == [0, 4, 4, 4, 4, 4]
    assert [len(p) for p in c["patterns"]] == [32, 50, 50, 50, 50, 50]


@pytest.mark.parametrize("name,fn", [("basic_test", "small_chrY_x100.fa"), ("easy", "easy.fa"),
                                     ("complex", "small_chrY_x100.fa")])
def test_batch_script_fixtures(name, fn):
    """tests/golden/expected/*.txt (the reference binary's result lines for the invocations of its
    scripts/basic_test.batch:10 and scripts/run_tests:31,56) == the oracle (banded, multi-threaded)."""
    with open(os.path.join(H.GOLDEN_DIR, "dna", fn), "rb") as f:
        text = f.read()
    pats, want = [], []
    with open(os.path.join(H.GOLDEN_DIR, "expected", name + ".txt"), "rb") as f:
        for line in f.read().splitlines():
            head, cnt = line.rsplit(b">: ", 1)
            pats.append(head[len(b"Number of matches for pattern <"):])
            want.append(int(cnt))
    assert len(pats) in (3, 6)
    assert H.oracle_counts(text, pats, 0, banded=True) == want


def test_single_thread_equals_multi_thread():
    c = next(c for c in CASES if c["name"] == "chrY_k3")
    text = H.case_text(c)
    for p in c["patterns"]:
        a = H.oracle().oracle_count(text, len(text), p, len(p), c["k"])
        b = H.oracle().oracle_count_range_mt(text, len(text), p, len(p), c["k"], 0, len(text), 3)
        assert a == b


def test_range_splits_add_up():
    """owner-computes partition: counts over [0,a)+[a,b)+[b,n) == whole."""
    c = next(c for c in CASES if c["name"] == "chrY_k2")
    text = H.case_text(c)
    n = len(text)
    for p, want in zip(c["patterns"], c["counts"]):
        parts = [H.oracle().oracle_count_range(text, n, p, len(p), c["k"], lo, hi)
                 for lo, hi in ((0, 400), (400, 1296), (1296, n))]
        assert sum(parts) == want


@pytest.mark.skipif(not os.path.exists(H.REF_UTILS_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_window_distance_equals_reference_function():
    """function-level pin: oracle_window_distance == the reference's levenshtein()
    compiled from /root/reference/src/utils.c into oracle/_ref/libref_utils.so."""
    ref = ctypes.CDLL(H.REF_UTILS_SO)
    ref.levenshtein.restype = ctypes.c_int
    ref.levenshtein.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    rnd = random.Random(7)
    for _ in range(3000):
        m = rnd.randint(1, 140)
        alpha = rnd.choice([b"ab", b"ACGT", bytes(range(1, 256))])
        p = bytes(rnd.choice(alpha) for _ in range(m))
        t = bytearray(p)
        for _e in range(rnd.randint(0, 6)):
            t[rnd.randrange(m)] = rnd.choice(alpha)
        if rnd.random() < 0.3:
            s = rnd.randint(1, 3)
            t = t[s:] + t[:s]
        t = bytes(t)
        col = (ctypes.c_int * (m + 1))()
        assert ref.levenshtein(p, t, m, col) == H.window_distance(p, t)
