#!/bin/bash
# tools/gpurun.sh <what> [args…] — the one script that runs on the GPU box (invoked as: gpurun -- 'bash tools/gpurun.sh <what> …').
# Everything it writes goes to gpurun_out/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
#   tests [pytest args]       pytest -m gpu (default: the whole GPU suite)
#   fast                      pytest -m 'gpu and not slow' (everything but the SD-size oracle / subprocess tests: 119 tests in 47 s)
#   smoke                     __graft_entry__.smoke()
#   bench [bench.py args]     python bench.py …                       -> gpurun_out/bench.json
#   trace <tag> [bench args]  rocprofv3 --kernel-trace of bench.py + per-symbol CSV (tools/prof_stats.py)
#                             -> gpurun_out/<tag>_bench_kernel_stats.csv   (the fp32 step alone: --no-config3 is implied)
#   trace3 <tag>              the same for the config-3 (bf16, bs=4) leg alone -> gpurun_out/<tag>_config3_kernel_stats.csv
#   pmc <tag> [bench args]    two PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with trace domains other than the kernel
#                             trace) of the fp32 leg + tools/pmc_traffic.py -> gpurun_out/<tag>_pmc_hbm_traffic.json
#   pmc3 <tag>                the same for the config-3 leg alone     -> gpurun_out/<tag>_pmc_hbm_traffic_config3.json
#   (every trace / PMC summary carries the stamp of the tree it was measured on: celebbasis_amd.build.source_digest())
#   pmcsq <tag> <script.py>   three --pmc passes (SQ busy / waits, instruction mix, L2 hit + TA busy) of a fixed workload
#   py <script> [args]        python <script> …                       -> gpurun_out/<script name>.log
# Box time measured in round 3 (what a call is charged, + ~20 s per call): full `tests` ~7 min (the CPU oracle at SD size is most of it); one
# bench.py process ~25 - 30 s; `trace` + `trace3` + `pmc` + `pmc3` (six rocprofv3 runs of bench.py) ~2 min (the call that ran suite + smoke + bench + all four was
# charged 550 s) — i.e. re-stamping profiles/ after a kernel change costs ~3 min with a bench run, a pmcsq of a micro-workload < 1 min, the CLI with --report (220 steps) ~25 s.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
what=${1:-tests}; shift || true
case "$what" in
  tests)
    if [ $# -eq 0 ]; then set -- tests -m gpu; fi
    ( time python -m pytest "$@" -q -p no:cacheprovider --no-header ) > gpurun_out/pytest.log 2>&1
    tail -25 gpurun_out/pytest.log ;;
  fast)
    ( time python -m pytest tests -m "gpu and not slow" -q -p no:cacheprovider --no-header --durations=10 "$@" ) > gpurun_out/pytest_fast.log 2>&1
    tail -25 gpurun_out/pytest_fast.log ;;
  smoke)
    python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log ;;
  bench)
    python bench.py "$@" > gpurun_out/bench.log 2>&1
    grep '^{' gpurun_out/bench.log | tail -1 > gpurun_out/bench.json; cut -c1-400 gpurun_out/bench.json ;;
  trace)
    tag=$1; shift
    STEPS=6
    cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/trace_$tag -o bench -- \
      python $REPO/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-pipelined --no-config3 --no-graph "$@" > $REPO/gpurun_out/trace_$tag.log 2>&1
    cd $REPO
    db=$(find gpurun_out/trace_$tag -name "*.db" | head -1)
    # every measurement leg of bench.py runs warmup + steps + 1 profiled step of its own kernels
    python tools/prof_stats.py "$db" gpurun_out/${tag}_bench_kernel_stats.csv $((STEPS + 3)) | head -60
    python tools/trace_overlap.py "$db" > gpurun_out/${tag}_stream_overlap.txt 2>&1; cat gpurun_out/${tag}_stream_overlap.txt
    python tools/trace_timeline.py "$db" > gpurun_out/${tag}_timeline.txt 2>&1; tail -15 gpurun_out/${tag}_timeline.txt
    find gpurun_out/trace_$tag -size +8M -delete ;;
  trace3)    # the config-3 (bf16, bs=4) leg alone -> gpurun_out/<tag>_config3_kernel_stats.csv
    tag=$1; shift
    STEPS=6
    cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/trace3_$tag -o bench -- \
      python $REPO/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-pipelined --no-graph --trace-leg config3 "$@" > $REPO/gpurun_out/trace3_$tag.log 2>&1
    cd $REPO
    db=$(find gpurun_out/trace3_$tag -name "*.db" | head -1)
    python tools/prof_stats.py "$db" gpurun_out/${tag}_config3_kernel_stats.csv $((STEPS + 3)) | head -40
    python tools/trace_overlap.py "$db" > gpurun_out/${tag}_config3_stream_overlap.txt 2>&1; cat gpurun_out/${tag}_config3_stream_overlap.txt
    python tools/trace_timeline.py "$db" > gpurun_out/${tag}_config3_timeline.txt 2>&1; tail -15 gpurun_out/${tag}_config3_timeline.txt
    find gpurun_out/trace3_$tag -size +8M -delete ;;
  pmc|pmc3)  # pmc = the fp32 (config-2) leg alone, pmc3 = the config-3 (bf16, bs=4) leg alone: per-symbol averages are not mixed
    tag=$1; shift
    if [ "$what" = pmc3 ]; then leg="--trace-leg config3"; sfx=_config3; else leg="--no-config3"; sfx=""; fi
    cd /tmp
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c -d $REPO/gpurun_out/pmc${sfx}_${tag}_$c -o p --output-format csv -- \
        python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-graph $leg "$@" > $REPO/gpurun_out/pmc${sfx}_${tag}_$c.log 2>&1
    done
    cd $REPO
    python tools/pmc_traffic.py gpurun_out/pmc${sfx}_${tag}_FETCH_SIZE gpurun_out/pmc${sfx}_${tag}_WRITE_SIZE gpurun_out/${tag}_pmc_hbm_traffic${sfx}.json
    find gpurun_out -name "*.csv" -size +8M -delete ;;
  pmcsq)     # SQ / LDS / cache counters of a fixed workload script: pmcsq <tag> <script.py>
    tag=$1; script=$2
    cd /tmp
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $REPO/gpurun_out/pmcsq_${tag}_a -o p --output-format csv -- python $REPO/$script > $REPO/gpurun_out/pmcsq_${tag}_a.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_WAIT_INST_LDS -d $REPO/gpurun_out/pmcsq_${tag}_b -o p --output-format csv -- python $REPO/$script > $REPO/gpurun_out/pmcsq_${tag}_b.log 2>&1
    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum TCC_REQ_sum -d $REPO/gpurun_out/pmcsq_${tag}_c -o p --output-format csv -- python $REPO/$script > $REPO/gpurun_out/pmcsq_${tag}_c.log 2>&1
    cd $REPO
    python tools/pmc_summary.py gpurun_out/pmcsq_${tag}_a gpurun_out/pmcsq_${tag}_b gpurun_out/pmcsq_${tag}_c > gpurun_out/pmcsq_${tag}.txt 2>&1
    find gpurun_out -name "*.csv" -size +8M -delete
    cat gpurun_out/pmcsq_${tag}.txt ;;
  py)
    script=$1; shift
    python "$script" "$@" > gpurun_out/$(basename "$script" .py).log 2>&1; tail -30 gpurun_out/$(basename "$script" .py).log ;;
  *) echo "unknown target $what"; exit 2 ;;
esac
