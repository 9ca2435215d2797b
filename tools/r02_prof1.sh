#!/bin/bash
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$PWD TMPDIR=/tmp
export FASTMC_LIB=$PWD/fast_amd/libfastmc_b3.so
rocprofv3 -L > gpurun_out/rocprof_counters_list.txt 2>&1
OUT=$PWD/gpurun_out/prof_r02a
mkdir -p $OUT
BENCH="python3 $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL --output-format csv -d "$OUT/pmc_sq3" -- $BENCH > "$OUT/pmc_sq3.log" 2>&1
python3 $PWD/tools/summarise_profile.py "$OUT" > "$OUT/summary.md" 2>&1
rm -rf $OUT/*/*/*.db 2>/dev/null
cat "$OUT/summary.md" | grep -v "^$" | grep -E "rows_wave|cols_wave|kernel \||PMC|stats" 
tail -3 $OUT/pmc_sq3.log
