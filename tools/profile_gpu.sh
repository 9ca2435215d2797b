#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py.
# Output under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
#   tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r02}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# The bench's OWN step: same launch sizes (library default batch), same pipeline mode (two steps in flight), default steps /
# warm-up, and the one-call pass behind pipeline.rows_ms -- only the second generator pass, the extras, the CPU baseline and the
# 5-second sustained run are left out.  The default generator is the float64 one (the headline); --rng-precision f32 profiles the
# opt-in float32 draw.  bench.py prints its launch plan (roofline.launch_plan), summarise_profile.py reads the per-dispatch
# times against it.
BENCH="python3 $PWD/bench.py --no-cpu-baseline --no-extras --no-sustained --no-f32-draw-pass $*"
# 1) kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
# 2) PMC passes (separate runs; FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
# the matrix pipe (expected idle -- no MFMA in any kernel, profiles/r06_ubench_mfma64_gate.txt says why)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_sq3" -- $BENCH > "$OUT/pmc_sq3.log" 2>&1
grep "^{\"metric\"" "$OUT/trace.log" > "$OUT/bench_line_under_rocprof.json"
NPX=1024; PRC=f64; A=("$@"); for i in "${!A[@]}"; do [ "${A[$i]}" == "config3" ] && NPX=2048; done
for i in "${!A[@]}"; do [ "${A[$i]}" == "--npxls" ] && NPX=${A[$((i+1))]}; [ "${A[$i]}" == "--precision" ] && PRC=${A[$((i+1))]}; done
python3 $PWD/tools/summarise_profile.py "$OUT" "$OUT/${TAG}_counters.json" $PRC $NPX > "$OUT/summary.md" 2>&1
rm -rf "$OUT"/*/*/*.db 2>/dev/null
cat "$OUT/summary.md"
