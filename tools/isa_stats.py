#!/usr/bin/env python3
"""Instruction mix of the hot kernels, counted from the gfx950 code hipcc generates (no GPU needed).

    python tools/isa_stats.py [--json fast_amd/kernel_isa_stats.json] [-DFLAG ...]

Compiles fast_amd/csrc/fastmc.hip to assembly (device only), cuts out each kernel named in KERNELS, finds its main
loop (the longest backward-branch span: for `k_rows_wave` one iteration = one spectrum row of one realisation, for
`k_cols_wave` the straight-line body = one window column) and counts the instructions of one pass through it by issue
class.  The stages of the transform are fully unrolled, so the static count of the loop body IS the dynamic count per
row, up to the exec-masked tails (`oi < Np`) which are counted as executed.

bench.py multiplies these counts by the rows it ran and by the per-class issue costs to report the share of the
SIMDs' issue slots the run used (`roofline.issue`), so that figure always belongs to the binary that ran:
`__graft_entry__.build()` regenerates the JSON next to libfastmc.so.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fast_amd", "csrc", "fastmc.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize", "-Wno-unused-result",
         "-Wno-unused-value", "-Wno-unused-command-line-argument", "-Wno-unused-function", "--cuda-device-only", "-S",
         "-DFMC_ISA_SUBSET"]       # only the priced kernels are instantiated: identical code for them, a tenth of the compile time

# demangled-name prefixes of the kernels whose counts bench.py uses
KERNELS = {
    # 1024^2 with a window of up to 96 pixels: the dense-image, sixteen-wave instantiations (last template argument 1)
    # (4: lanes factored 16 x 4, six of sixteen exchange-2 planes: the default for the centred BASELINE window)
    "rows_f64_1024": "void fmc::k_rows_wave<double, 16, 2, 0, 1, 4>(",
    "cols_f64_1024": "void fmc::k_cols_wave<double, 16, 2, 0, 1, 4>(",
    "rows_f32_1024": "void fmc::k_rows_wave<float, 16, 2, 0, 1, 4>(",
    "cols_f32_1024": "void fmc::k_cols_wave<float, 16, 2, 0, 1, 4>(",
    "rows_f64_1024_12waves": "void fmc::k_rows_wave<double, 16, 2, 0, 1, 5>(",
    # 2048^2 (round 6): the packed sub-rows with a run-time count -- eight sub-rows of 256 points, FOUR rows per wavefront
    "rows_f64_2048": "void fmc::k_rows_pks<double, 1, -2, 0>(",
    # (its column kernel k_cols_pks<double, 1, -2, 0> holds a loop of eight passes and a rolled detector loop: a static count of the
    # body is not a count per column, and nothing prices the column pass by instructions -- it is not listed)
    # the float64 generator fused into the row (MODE 2; round 4)
    "rows_f64_1024_gen64": "void fmc::k_rows_wave<double, 16, 2, 2, 1, 4>(",
    "rows_f64_2048_gen64": "void fmc::k_rows_pks<double, 1, -2, 2>(",
}
# sub-rows per row of the split-row kernels: their row loop contains the sub-row loop, counted as outer + S x inner
SUB_ROWS = {"rows_f64_2048": 8, "rows_f64_2048_gen64": 8}
# rows one pass of the row loop transforms (packed sub-rows: the G = 4 rows of a unit): the counts are divided by it
ROWS_PER_PASS = {"rows_f64_2048": 4, "rows_f64_2048_gen64": 4}

# --packed: the packed rows of the small grids (translation unit 10; unit = G rows / columns of one wavefront)
PACKED_KERNELS = {
    "rows_f64_128_packed": "void fmc::k_rows_pk<double, 0, 0, 0>(",
    "cols_f64_128_packed": "void fmc::k_cols_pk<double, 0, 0, 0>(",
    "rows_f64_256_packed": "void fmc::k_rows_pk<double, 1, 0, 0>(",
    "cols_f64_256_packed": "void fmc::k_cols_pk<double, 1, 0, 0>(",
    "rows_f64_512_packed": "void fmc::k_rows_pk<double, 2, 0, 0>(",
    "cols_f64_512_packed": "void fmc::k_cols_pk<double, 2, 0, 0>(",
    # the float64 generator fused into the packed rows (MODE 2)
    "rows_f64_128_packed_gen64": "void fmc::k_rows_pk<double, 0, 2, 0>(",
    "rows_f64_256_packed_gen64": "void fmc::k_rows_pk<double, 1, 2, 0>(",
    "rows_f64_512_packed_gen64": "void fmc::k_rows_pk<double, 2, 2, 0>(",
}

TRANS = re.compile(r"^v_(log|sqrt|sin|cos|exp|rcp|rsq)_(f32|f16|f64)")


def classify(m):
    if m.startswith("v_"):
        if m.startswith("v_cvt") and "f64" in m:
            return "valu_cvt_f64"
        if re.search(r"_f64(_e32|_e64|_dpp)?$", m):
            return "valu_f64"
        if TRANS.match(m):
            return "valu_trans"
        if m.startswith(("v_mad_u64", "v_mul_lo_u32", "v_mul_hi")):
            return "valu_int_quarter"          # quarter-rate integer multiplies
        return "valu_other"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if m.startswith("s_waitcnt"):
        return "s_waitcnt"
    if m.startswith("s_"):
        return "salu"
    return "other"


def flop_f64(m):
    """float64 flops per lane of one instruction."""
    if not re.search(r"_f64(_e32|_e64|_dpp)?$", m) or m.startswith("v_cvt"):
        return 0
    return 2 if m.startswith(("v_fma", "v_fmac")) else 1


def flop_f32(m):
    """float32 add / mul / fma flops per lane of one instruction (packed forms count both halves)."""
    mm = re.match(r"^v_(pk_)?(add|sub|subrev|mul|fma|fmac|fmamk|fmaak|mac|mad)_(f32)", m)
    if not mm:
        return 0
    f = 2 if mm.group(2) in ("fma", "fmac", "fmamk", "fmaak", "mac", "mad") else 1
    return f * (2 if mm.group(1) else 1)


def kernel_bodies(asm):
    """{mangled name: [instruction lines]} for every function of the assembly text."""
    out, cur, name = {}, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is not None:
            cur.append(line)
            if line.strip().startswith("s_endpgm"):
                cur = None
    return out


def main_loop(lines):
    """(start, end) line indices of the longest backward-branch span, or the whole body when there is none."""
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\w+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if span[0] < 16 and span[1] > len(lines) - 16:
                continue      # a branch from the kernel's last block back over its whole body (a cold block placed last): not a loop
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    return best if best and best[1] - best[0] > 200 else (0, len(lines) - 1)


def inner_loop(lines, a, b):
    """(start, end) of the longest backward-branch span strictly inside (a, b), or None: the sub-row loop of a split row."""
    labels = {}
    for i in range(a, b + 1):
        m = re.match(r"^(\.LBB\w+):", lines[i])
        if m:
            labels[m.group(1)] = i
    best = None
    for i in range(a, b):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\w+)", lines[i])
        if m and m.group(1) in labels and a < labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    return best if best and best[1] - best[0] > 200 else None


def n_instr(lines):
    return sum(1 for l in lines if re.match(r"^\s+[a-z]", l) and not l.strip().startswith((";", ".", "//")))


def count(lines):
    mix, by_class, flops, flops32 = collections.Counter(), collections.Counter(), 0, 0
    for l in lines:
        l = l.strip()
        if not l or l.startswith((";", ".", "//")):
            continue
        m = l.split()[0]
        if not re.match(r"^[a-z]", m):
            continue
        mix[m] += 1
        by_class[classify(m)] += 1
        flops += flop_f64(m)
        flops32 += flop_f32(m)
    return mix, by_class, flops, flops32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None, help="write the per-kernel counts here")
    ap.add_argument("--asm", default=None, help="reuse this assembly file instead of compiling")
    ap.add_argument("--top", type=int, default=0, help="print the N most frequent mnemonics per kernel")
    ap.add_argument("--packed", action="store_true", help="the packed-row kernels of 128 / 256 / 512 instead (column kernels: static "
                    "count of the whole body, which includes the never-taken library fall-backs of sincos and both sub-harmonic branches)")
    args, extra = ap.parse_known_args()
    kernels, flags = KERNELS, FLAGS
    if args.packed:
        kernels = PACKED_KERNELS
        flags = [f for f in FLAGS if f != "-DFMC_ISA_SUBSET"] + ["-DFMC_SPLIT_BUILD", "-DFMC_TU=10", "-DFMC_ONLY_F64"]
    if args.asm:
        asm = open(args.asm).read()
    else:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "fmc.s")
            r = subprocess.run([HIPCC] + flags + extra + ["-o", out, SRC], capture_output=True, text=True)
            if r.returncode != 0:
                sys.exit(r.stderr[-4000:])
            asm = open(out).read()
    bodies = kernel_bodies(asm)
    names = list(bodies)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    result = {}
    for tag, prefix in kernels.items():
        hit = [n for n, d in zip(names, dem) if d.startswith(prefix)]
        if not hit:
            continue
        lines = bodies[hit[0]]
        a, b = main_loop(lines)
        if tag.startswith("rows"):
            # k_rows_wave walks tiles (round 5: the workgroups of a large launch stay): the tile loop holds the row loop plus a
            # few dozen instructions of index arithmetic and a barrier per tile of 8 rows per wave -- the row loop is the unit
            row = inner_loop(lines, a, b)
            if row and n_instr(lines[a:row[0]] + lines[row[1] + 1:b + 1]) <= 80:
                a, b = row
        if tag.startswith("cols"):
            a, b = 0, len(lines) - 1          # one column per wave: the whole body (its small loops are the table load)
        S = SUB_ROWS.get(tag, 1)
        inner = inner_loop(lines, a, b) if S > 1 else None
        if inner:
            # one row = the row loop's own instructions + S passes through the sub-row loop
            ia, ib = inner
            mo, co, fo, f32o = count(lines[a:ia] + lines[ib + 1:b + 1])
            mi, ci, fi, f32i = count(lines[ia:ib + 1])
            mix = mo + collections.Counter({k: S * v for k, v in mi.items()})
            by_class = co + collections.Counter({k: S * v for k, v in ci.items()})
            flops, flops32 = fo + S * fi, f32o + S * f32i
        else:
            mix, by_class, flops, flops32 = count(lines[a:b + 1])
        G = ROWS_PER_PASS.get(tag, 1)
        if G > 1:
            by_class = collections.Counter({k: v / G for k, v in by_class.items()})
            flops, flops32 = flops / G, flops32 / G
        valu = sum(v for k, v in by_class.items() if k.startswith("valu"))
        result[tag] = {"kernel": prefix.rstrip("(").replace("void fmc::", ""), "unit": "one row (rows) / one column (cols) per wave",
                       "instructions": dict(by_class), "valu_total": valu, "f64_flop_per_lane": flops, "f32_flop_per_lane": flops32,
                       "loop_lines": [a, b], "body_lines": len(lines)}
        if inner:
            result[tag]["sub_rows"] = S
            result[tag]["rows_per_pass"] = G
            result[tag]["sub_row_loop_lines"] = list(inner)
        print(f"{tag:16s} VALU {valu:5g}  " + "  ".join(f"{k} {v:g}" for k, v in sorted(by_class.items())) + f"  f64 flop/lane {flops:g}")
        if args.top:
            print("   ", ", ".join(f"{m} {n}" for m, n in mix.most_common(args.top)))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(result, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
