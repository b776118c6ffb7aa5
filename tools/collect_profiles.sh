#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 evidence for the bench's dominant kernel. Outputs under gpurun_out/prof_final/.
# gpurun MERGES the outputs into the local gpurun_out/: remove the local gpurun_out/prof_final first (or rely on
# summarize_profiles.py picking the newest file of each kind).
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip tracing).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_final
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1   # first-import / allocator warm-up outside the profiles
python3 bench.py --mode train --steps 2 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_infer -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_infer.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python3 bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_train.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
# the same three passes over the training step (forward cells + BPTT kernels)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_train_fetch -- python3 bench.py --mode train --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_train_write -- python3 bench.py --mode train --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_train_sq -- python3 bench.py --mode train --steps 2 --warmup 1 --prewarm 0 --no-cpu-baseline --no-extras > /dev/null 2>&1
# extra configurations (kernel stats only): exact-fp32 operands, PredRNN forward / training step
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_infer_f32 -- python3 bench.py --precision f32 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_infer_f32.log 2>&1
python3 bench.py --model predrnn-pp --mode train --steps 1 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_predrnn_infer -- python3 bench.py --model predrnn-pp --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_predrnn_infer.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_predrnn_train -- python3 bench.py --model predrnn-pp --mode train --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_predrnn_train.log 2>&1
# keep only the small summaries (kernel_trace.csv of the bench runs is large)
find $OUT -name "*kernel_trace.csv" -path "*trace_*" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
