#!/usr/bin/env python3
"""Generate tests/golden/golden.json by running the REFERENCE binary.

Runs oracle/_ref/apm_sequential (built by oracle/Makefile from
/root/reference/src/{utils,sequential}.c, the sources staying where they lie)
on the reference's own data files (copied as data into tests/golden/dna/) and
on small synthetic texts, and records `Number of matches` per pattern
(stdout contract: /root/reference/src/sequential.c:157-160).

Only runs in the build container (needs oracle/_ref).  The committed JSON plus
the data files are what travels; the reference sources never do.
"""
import base64
import hashlib
import json
import os
import random
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "apm_sequential")
REF_DNA = "/root/reference/dna"
GOLD = os.path.join(ROOT, "tests", "golden")
DNA = os.path.join(GOLD, "dna")



def run_ref(k, text_path, patterns):
    out = subprocess.run([REF_BIN, str(k), text_path] + patterns,
                         capture_output=True, check=True).stdout
    counts = []
    pos = out.index(b"Number of matches")
    for p in patterns:      # patterns may contain '\n': walk the output sequentially
        head = b"Number of matches for pattern <" + p + b">: "
        assert out.startswith(head, pos), (out, pos)
        end = out.index(b"\n", pos + len(head))
        counts.append(int(out[pos + len(head):end]))
        pos = end + 1
    assert pos == len(out), out
    return counts


def write_stdout_fixtures():
    """tests/golden/expected/*.txt: the `Number of matches` lines the reference prints for the
    invocations of its own batch scripts (scripts/basic_test.batch:10, scripts/run_tests:31,56) --
    what scripts/run_tests.sh diffs the GPU CLI against."""
    out_dir = os.path.join(GOLD, "expected")
    os.makedirs(out_dir, exist_ok=True)
    L = {n: read("line_%s.fa" % n).split()[0] for n in ("10", "20", "20783", "non_existent")}
    runs = {
        "basic_test": ("small_chrY_x100.fa", [L["non_existent"]] + [L["20783"]] * 5),
        "easy": ("easy.fa", ["123", "456", "78934"]),
        "complex": ("small_chrY_x100.fa", [L["10"], L["20"], L["non_existent"]] * 2),
    }
    for name, (fn, pats) in runs.items():
        r = subprocess.run([REF_BIN, "0", os.path.join(DNA, fn)] + pats, capture_output=True, check=True)
        lines = [l for l in r.stdout.decode("latin-1").splitlines() if l.startswith("Number of matches")]
        with open(os.path.join(out_dir, name + ".txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    print("wrote %d stdout fixtures" % len(runs))


def read(name):
    with open(os.path.join(DNA, name), "rb") as f:
        return f.read().decode("latin-1")


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("oracle/_ref/apm_sequential missing: run `make -C oracle` first")
    os.makedirs(DNA, exist_ok=True)
    for fn in sorted(os.listdir(REF_DNA)):        # data files, not source
        shutil.copyfile(os.path.join(REF_DNA, fn), os.path.join(DNA, fn))

    L = {n: read("line_%s.fa" % n) for n in ("10", "20", "5", "1131", "20783", "non_existent")}
    chrY_flat = read("small_chrY.fa").replace("\n", "")
    probe16 = "AGAAGAGCACCTGGTT"

    cases = []  # dict(name, file | text, k, patterns)

    def add_file(name, fn, k, pats):
        cases.append(dict(name=name, file=fn, k=k, patterns=pats))

    def add_text(name, text, k, pats):
        cases.append(dict(name=name, text=text, k=k, patterns=pats))

    # --- the reference's own canonical invocations ---
    # scripts/basic_test.batch:10 == BASELINE config #1, README.md:58-63
    add_file("cfg1_basic_test", "small_chrY_x100.fa", 0,
             [L["non_existent"]] + [L["20783"]] * 5)
    # scripts/run_tests:31 and :52
    add_file("run_tests_easy", "easy.fa", 0, ["123", "456", "78934"])
    add_file("run_tests_complex", "small_chrY_x100.fa", 0,
             [L["10"], L["20"], L["non_existent"]] * 2)

    # --- k sweeps on the reference's data files ---
    five = [L["10"], L["20"], L["1131"], L["non_existent"], probe16]
    for k in (0, 1, 2, 3, 4, 5):
        add_file("x100_k%d" % k, "small_chrY_x100.fa", k, five)
    seven = [L["10"], L["20"], L["5"], L["1131"], L["non_existent"], probe16, "GGGG"]
    for k in (0, 1, 2, 3, 4, 5, 6, 7):
        add_file("chrY_k%d" % k, "small_chrY.fa", k, seven)
    add_file("medium_k0", "small_chrY_medium.fa", 0, [L["10"]])
    add_file("medium_k2", "small_chrY_medium.fa", 2, [L["10"], probe16])
    add_file("bigger_k0", "small_chrY_bigger.fa", 0, [L["10"], L["20"], L["5"], L["1131"]])
    # m sweep: slices of the newline-stripped small_chrY
    for k in (2, 3):
        add_file("x100_msweep_k%d" % k, "small_chrY_x100.fa", k,
                 [chrY_flat[100:100 + m] for m in (1, 2, 3, 8, 16, 31, 32, 33, 50, 63, 64, 65, 96, 127, 128)])
    add_file("x100_long_patterns_k3", "small_chrY_x100.fa", 3,
             [chrY_flat[40:40 + m] for m in (129, 160, 200, 300)])
    add_file("chrY_long_patterns_k5", "small_chrY.fa", 5,
             [chrY_flat[40:40 + m] for m in (129, 200, 257, 600, 1300)])

    # --- hand-made edge cases (truncated tails, m > n, n <= k, k >= m, newline) ---
    for k in (0, 1, 2):
        add_text("A16_k%d" % k, "A" * 16, k, ["AAAB", "AAAA", "A", "B", "AAAAAAAAAAAAAAAA", "AAAAAAAAAAAAAAAAA"])
    for k in (0, 1, 3, 9, 10, 12, 20):
        add_text("m_gt_n_k%d" % k, "ACGTACGTAC", k, ["ACGTACGTACGT", "ACGTACGTACTT", "ACGT", "TTTTTTTTTTTTTTTTTTTT"])
    for k in (0, 1):
        add_text("newline_k%d" % k, "ACGT\nACGT\n", k, ["ACGT", "GTAC", "T\nA", "\n"])
    add_text("single_byte_text", "A", 0, ["A", "C", "AA"])
    add_text("single_byte_text_k1", "A", 1, ["A", "C", "AA"])
    add_text("k_ge_m", "ACGTTGCAACGTTGCA" * 4, 6, ["ACGT", "ACGTTG", "ACGTTGC", "GGGGGGG"])

    # --- seeded random texts: small alphabets so that matches are common ---
    rnd = random.Random(560)
    alphabets = {"ab": "ab", "acgt": "ACGT", "abc_nl": "abc\n", "bytes": "".join(chr(c) for c in range(1, 256))}
    for aname, alpha in alphabets.items():
        for trial in range(3):
            n = rnd.choice([1, 7, 63, 64, 65, 300, 1000, 2049])
            text = "".join(rnd.choice(alpha) for _ in range(n))
            pats = []
            for _ in range(6):
                m = rnd.choice([1, 2, 3, 5, 8, 16, 17, 31, 32, 33, 40, 64, 65, 100, 128])
                if rnd.random() < 0.7 and n > m:
                    o = rnd.randrange(0, n - m + 1)
                    p = list(text[o:o + m])
                    for _e in range(rnd.randrange(0, 4)):      # mutate
                        p[rnd.randrange(m)] = rnd.choice(alpha)
                    p = "".join(p)
                else:
                    p = "".join(rnd.choice(alpha) for _ in range(m))
                pats.append(p)
            k = rnd.choice([0, 1, 2, 3, 4, 5])
            add_text("rand_%s_%d" % (aname, trial), text, k, pats)
    # long random DNA with planted edits, several k, pattern lengths of the BASELINE configs
    text = "".join(rnd.choice("ACGT") for _ in range(20000))
    for k in (0, 1, 2, 3, 5):
        pats = []
        for m in (16, 32, 50, 64, 100, 128):
            o = rnd.randrange(0, len(text) - m)
            p = list(text[o:o + m])
            for _e in range(rnd.randrange(0, k + 2)):
                r = rnd.random()
                pos = rnd.randrange(len(p))
                if r < 0.5:
                    p[pos] = rnd.choice("ACGT")
                elif r < 0.75:
                    del p[pos]
                    p.append(rnd.choice("ACGT"))
                else:
                    p.insert(pos, rnd.choice("ACGT"))
                    p.pop()
            pats.append("".join(p))
        add_text("dna20k_k%d" % k, text, k, pats)

    # long, loose patterns (128 < m <= 256 with k > 7 or pieces shorter than 4 bytes: AUTO's WAVEFRONT route)
    add_file("chrY_loose_long_k60", "small_chrY.fa", 60, [chrY_flat[300:300 + m] for m in (150, 200, 256)])
    add_file("chrY_loose_long_k8", "small_chrY.fa", 8, [chrY_flat[77:77 + m] for m in (129, 200, 256)])
    pats = []
    for m in (129, 200, 256):
        o = rnd.randrange(0, len(text) - m)
        p = list(text[o:o + m])
        for _e in range(9):
            r, pos = rnd.random(), rnd.randrange(len(p))
            if r < 0.5:
                p[pos] = rnd.choice("ACGT")
            elif r < 0.75:
                del p[pos]
                p.append(rnd.choice("ACGT"))
            else:
                p.insert(pos, rnd.choice("ACGT"))
                p.pop()
        pats.append("".join(p))
    add_text("dna20k_loose_long_k9", text, 9, pats)

    tmpdir = tempfile.mkdtemp(prefix="apm_golden_")

    def solve(case):
        if "file" in case:
            path = os.path.join(DNA, case["file"])
        else:
            path = os.path.join(tmpdir, case["name"] + ".txt")
            with open(path, "wb") as f:
                f.write(case["text"].encode("latin-1"))
        pats = [p.encode("latin-1") for p in case["patterns"]]
        case["counts"] = run_ref(case["k"], path, pats)
        with open(path, "rb") as f:
            case["text_sha256"] = hashlib.sha256(f.read()).hexdigest()
        return case

    with ThreadPoolExecutor(max_workers=8) as ex:
        done = list(ex.map(solve, cases))
    shutil.rmtree(tmpdir)

    for c in done:   # JSON-safe: latin-1 bytes as base64
        if "text" in c:
            c["text_b64"] = base64.b64encode(c.pop("text").encode("latin-1")).decode()
        c["patterns_b64"] = [base64.b64encode(p.encode("latin-1")).decode() for p in c.pop("patterns")]

    # CLI error-path vectors: rc + first stderr/stdout line
    def cli(args):
        r = subprocess.run([REF_BIN] + args, capture_output=True)
        return dict(args=args, rc=r.returncode,
                    stdout=r.stdout.decode("latin-1").replace(REF_BIN, "<exe>"),
                    stderr=r.stderr.decode("latin-1"))
    errors = [cli([]), cli(["0", "/nonexistent/file.fa", "ACGT"]),
              cli(["0", os.path.join(DNA, "easy.fa"), ""])]
    for e in errors:
        e["args"] = [a.replace(DNA, "<dna>") for a in e["args"]]
        e["stdout"] = e["stdout"].replace(DNA, "<dna>")

    out = dict(generator="oracle/gen_golden.py", reference_binary="oracle/_ref/apm_sequential "
               "(gcc -O3 -w, /root/reference/src/{utils,sequential}.c)",
               cases=done, cli_errors=errors)
    with open(os.path.join(GOLD, "golden.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote %d cases" % len(done))


if __name__ == "__main__":
    if "--stdout-fixtures" in sys.argv:  # only the small text fixtures, golden.json untouched
        write_stdout_fixtures()
    else:
        main()
        write_stdout_fixtures()
