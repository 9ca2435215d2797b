#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/profile_gpu.sh) into one markdown summary."""
import csv, glob, os, sys, collections

out = sys.argv[1]

def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))

def short(name):
    n = name.replace("void fmc::", "").replace("fmc::", "")
    return n.split("(")[0][:60]

print(f"# rocprofv3 summary ({os.path.basename(out)})\n")
for f in find("trace", "*kernel_stats.csv"):
    print("## kernel stats (rocprofv3 --kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
              f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    print()
def per_launch_table():
    """The row kernel's dispatches of the trace run, one by one, against bench.py's launch plan (roofline.launch_plan of the line
    the profiled command printed): which call each dispatch belongs to and how many realisations it held -- so that an average
    is only ever taken over EQUAL launches."""
    import json
    try:
        line = json.loads(open(os.path.join(out, "bench_line_under_rocprof.json")).read())
        plan = line["launch_plan"]
    except Exception as e:
        print(f"(no launch plan in the profiled bench line: {e})\n")
        return
    kernel, B = plan["kernel"], int(plan["batch"])
    seq = []                                  # (phase, realisations) per dispatch, in issue order
    for phase, n, count in plan["calls"]:
        one = [B] * (n // B) + ([n % B] if n % B else [])
        seq += [(phase, r) for _ in range(int(count)) for r in one]
    files = find("trace", "*kernel_trace.csv")
    if not files:
        return
    disp = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(files[0]))
                   if short(r["Kernel_Name"]) == short(kernel)), key=lambda t: t[0])
    N = int(line["metric"].split("(")[1].split("^")[0])
    ghz = (line.get("clock") or {}).get("effective_GHz")
    steps = int(line["steps"])
    print(f"## the row kernel launch by launch ({short(kernel)}, batch {B}, N = {N}; clock stamped in this run: {ghz and round(ghz, 3)} GHz)\n")
    if len(disp) != len(seq):
        print(f"(the trace holds {len(disp)} dispatches of this kernel, the plan {len(seq)}: not matched one to one)\n")
        return
    groups = collections.OrderedDict()
    for (phase, r), (t0, t1) in zip(seq, disp):
        groups.setdefault((phase, r), []).append((t1 - t0) / 1e3)
    print("| call | realisations in the launch | launches | mean us | min us | max us | us per realisation | cycles per row and SIMD at the stamped clock |")
    print("|---|---|---|---|---|---|---|---|")
    for (phase, r), us in groups.items():
        m = sum(us) / len(us)
        # a launch of r realisations is r N rows over 1024 SIMDs: cycles per row = time x clock x 1024 / (r N)
        cyc = m * 1e-6 * ghz * 1e9 * 1024 / (r * N) if ghz else float("nan")
        print(f"| {phase} | {r} | {len(us)} | {m:.1f} | {min(us):.1f} | {max(us):.1f} | {m / r:.4f} | {cyc:.0f} |")
    tot = collections.defaultdict(float)
    for (phase, r), us in groups.items():
        tot[phase] += sum(us)
    print()
    if "timed" in tot:
        print(f"rows per step, timed steps (sum of their launches / {steps} steps): **{tot['timed'] / steps / 1e3:.3f} ms**")
    if "one_call" in tot:
        print(f"rows per step, one-call pass (the same realisations as ONE call / {steps}): **{tot['one_call'] / steps / 1e3:.3f} ms**")
    pl = line.get("pipeline", {})
    print(f"the bench line of this very run: pipeline.rows_ms = {pl.get('rows_ms'):.3f} ms ({pl.get('kernel_ms_from')}; HIP events), "
          f"timed steps' event sum {pl.get('rows_ms_timed_steps_event_sum'):.3f} ms, ms_per_step {line['ms_per_step']:.3f}, "
          f"value {line['value']:.0f} it/s under the profiler\n")


per_launch_table()
for f in find("trace", "*kernel_trace.csv")[:1]:
    rows = list(csv.DictReader(open(f)))
    seen = {}
    for r in rows:
        k = short(r["Kernel_Name"])
        if k not in seen:
            seen[k] = r
    print("## launch geometry / resources (first dispatch of each kernel)\n")
    print("| kernel | grid | workgroup | VGPR | accum VGPR | SGPR | LDS B | scratch B |")
    print("|---|---|---|---|---|---|---|---|")
    for k, r in seen.items():
        print(f"| {k} | {r.get('Grid_Size_X','?')} | {r.get('Workgroup_Size_X','?')} | {r.get('VGPR_Count','?')} | "
              f"{r.get('Accum_VGPR_Count','?')} | {r.get('SGPR_Count','?')} | {r.get('LDS_Block_Size','?')} | {r.get('Scratch_Size','?')} |")
    print()
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_sq4"):  # (pmc_sq3: the matrix-pipe counters)
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    print(f"## PMC pass {sub} (per-dispatch averages)\n")
    names = sorted({c for k in agg for c in agg[k]})
    print("| kernel | dispatches | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    for k in agg:
        n = max(len(cnt[k]), 1)
        print(f"| {k} | {n} | " + " | ".join(f"{agg[k][c]/n:.4g}" for c in names) + " |")
    print()

# ---- machine-readable counters of the two hot kernels (bench.py reads profiles/latest_counters.json)
import json
import re


def _avg(sub, counter, prefix):
    tot, ids = 0.0, set()
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if short(r["Kernel_Name"]).startswith(prefix) and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"])
                ids.add(r["Dispatch_Id"])
    return tot / len(ids) if ids else None


def _kernel_ms(prefix):
    for f in find("trace", "*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if short(r["Name"]).startswith(prefix):
                return float(r["AverageNs"]) / 1e6, short(r["Name"])
    return None, None


if len(sys.argv) > 2:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    isa = {}
    try:
        isa = json.load(open(os.path.join(root, "fast_amd", "kernel_isa_stats.json")))
    except Exception:
        pass
    prec, npx = sys.argv[3] if len(sys.argv) > 3 else "f64", int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    doc = {"source": f"profiles/{os.path.basename(sys.argv[2]).replace('_counters.json', '')}_* (tools/profile_gpu.sh: rocprofv3 --kernel-trace --stats, "
                     "separate --pmc passes FETCH_SIZE / WRITE_SIZE / SQ; per-dispatch averages)",
           "precision": prec, "npxls": npx,
           "rows_valu_instructions_per_row": None,
           "fetch_correction": "x2 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section)"}
    # the row / column kernels of the profiled grid: the packed rows on 128 / 256 / 512, else the one-row-per-wave kernels
    # (... the packed sub-rows k_rows_pks / k_cols_pks share the prefix; the packed chirp-z rows k_rows_pbz; else the one-row-per-wave kernels)
    rows_k = next((k for k in ("k_rows_pk", "k_rows_pbz", "k_rows_blu", "k_rows_mr") if _kernel_ms(k)[0]), "k_rows_wave")
    cols_k = next((k for k in ("k_cols_pk", "k_cols_pbz", "k_cols_blu", "k_cols_mr") if _kernel_ms(k)[0]), "k_cols_wave")
    for tag, prefix in (("rows", rows_k), ("cols", cols_k)):
        ms, name = _kernel_ms(prefix)
        fetch, write = _avg("pmc_fetch", "FETCH_SIZE", prefix), _avg("pmc_write", "WRITE_SIZE", prefix)
        ent = {"kernel": name, "avg_launch_ms": ms, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write}
        if fetch is not None and write is not None:
            ent["hbm_bytes_per_launch"] = (2 * fetch + write) * 1024
        doc[tag] = ent
    # the instruction counts of exactly the row kernel that was profiled (bench.py matches counters to a run by this name)
    doc["rows_kernel"] = (doc["rows"].get("kernel") or "").strip()
    doc["cols_kernel"] = (doc["cols"].get("kernel") or "").strip()
    st = next((v for v in isa.values() if isinstance(v, dict) and v.get("kernel") == doc["rows_kernel"]), None)
    if st:
        doc["rows_valu_instructions_per_row"] = st["valu_total"]
    grbm = _avg("pmc_sq2", "GRBM_GUI_ACTIVE", rows_k)
    if grbm:
        simd_cycles = grbm / 8 * 1024          # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
        for key, sub, counter in (("valu_busy", "pmc_sq", "SQ_ACTIVE_INST_VALU"), ("issue_busy", "pmc_sq2", "SQ_ACTIVE_INST_ANY"),
                                  ("lds_issue_busy", "pmc_sq2", "SQ_ACTIVE_INST_LDS")):
            v = _avg(sub, counter, rows_k)
            if v:
                doc[key + "_counter_ratio"] = v * 4 / simd_cycles        # SQ_ACTIVE_INST_* count quad-cycles; NOT clamped
        # An any-instruction ratio above 1 cannot be a utilisation: it says the denominator (GRBM_GUI_ACTIVE / 8 x 1024 SIMD-cycles)
        # is that much too short.  Dividing the class ratios by it gives upper bounds of the true busy fractions.
        any_ratio = doc.get("issue_busy_counter_ratio")
        short_by = max(any_ratio, 1.0) if any_ratio else 1.0
        for key in ("valu_busy", "issue_busy", "lds_issue_busy"):
            if key + "_counter_ratio" in doc:
                doc[key] = doc[key + "_counter_ratio"] / short_by
        doc["busy_note"] = ("row kernel: *_counter_ratio = SQ_ACTIVE_INST_* x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), unclamped; an "
                            "any-instruction ratio above 1 means the denominator is that much too short, so valu_busy / issue_busy / "
                            "lds_issue_busy are the counter ratios divided by max(1, any-instruction ratio): upper bounds of the true fractions")
        doc["clock_GHz_profiled"] = grbm / 8 / (doc["rows"]["avg_launch_ms"] * 1e-3) / 1e9 if doc["rows"]["avg_launch_ms"] else None
    try:       # launch size of the profiled command: the per-launch byte counts scale with it
        line = json.loads(open(os.path.join(out, "bench_line_under_rocprof.json")).read())
        doc["realisations_per_launch"] = line["roofline"]["realisations_per_launch"]
    except Exception:
        doc["realisations_per_launch"] = None
    json.dump(doc, open(sys.argv[2], "w"), indent=1)
