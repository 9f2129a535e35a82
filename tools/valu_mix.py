"""Instruction-class mix of the full-DP kernels' loops, from the gfx950 assembly (build container, no GPU):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o kernels.s csrc/apm_kernels.hip
    python tools/valu_mix.py kernels.s apm_wavefront_kernel apm_bitpar_kernel
For every kernel named: the VALU instructions inside loops (between a backward branch and its target), counted once per
nesting level they sit in (an instruction of an inner loop weighs more than one of the outer loop around it), split into
the two issue classes tools/valu_probe.hip measured on MI355X (profiles/r02/valu_probe.txt): "simple" = a wave64
instruction every 2 cycles per SIMD, everything else 4.  bench.py prices the kernels' measured VALU instruction rate
against the ceiling of THAT mix:  1 / (f2 / peak2 + f4 / peak4)."""
import re, sys
SIMPLE = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32",
          "v_bitop3_b32")   # exactly the ops the probe measured at 2 cycles; v_add_co / v_cmp / v_cndmask measured at 4 or worse
src = open(sys.argv[1]).read().split("\n")
for kname in sys.argv[2:]:
    for start, line in enumerate(src):
        m = re.match(r"^(_Z\w*%s\w*):" % kname, line)
        if not m:
            continue
        name = m.group(1)
        body = []
        for l in src[start + 1:]:
            if l.strip().startswith("s_endpgm"):
                break
            body.append(l)
        labels = {}
        for i, l in enumerate(body):
            mm = re.match(r"^(\.LBB\w+):", l)
            if mm:
                labels[mm.group(1)] = i
        depth = [0] * len(body)
        for i, l in enumerate(body):
            mm = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\w+)|^\s+s_branch\s+(\.LBB\w+)", l)
            if mm:
                t = labels.get(mm.group(1) or mm.group(2))
                if t is not None and t <= i:                   # backward branch: a loop [t, i]
                    for j in range(t, i + 1):
                        depth[j] += 1
        n2 = n4 = 0
        dpp = 0
        for i, l in enumerate(body):
            op = l.strip().split(" ")[0]
            if not op.startswith("v_") or depth[i] == 0:
                continue
            w = 4 ** (depth[i] - 1)                             # an inner loop runs many times per trip of the outer one
            is_dpp = "dpp" in l or "row_" in l or "wave_sh" in l
            simple = any(op.startswith(s) for s in SIMPLE) and not is_dpp and not op.startswith("v_pk_")
            if simple:
                n2 += w
            else:
                n4 += w
            dpp += w if is_dpp else 0
        tot = n2 + n4
        if tot:
            print("%s  loop VALU (weighted) %d: 2-cycle class %.3f, 4-cycle class %.3f (DPP %.3f)" % (name, tot, n2 / tot, n4 / tot, dpp / tot))
