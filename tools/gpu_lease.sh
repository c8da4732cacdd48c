#!/bin/bash
# ONE parameterised lease script (the one-off tools/gpu_r3_* / gpu_r4_* scripts are gone).
#   gpurun -- bash tools/gpu_lease.sh TAG STEP [STEP ...]        -> gpurun_out/TAG/
# steps:
#   tests         pytest -m gpu (whole suite)
#   tests:EXPR    pytest -m gpu -k EXPR
#   smoke         __graft_entry__.smoke()
#   bench         bench.py, default flags          -> bench.json
#   chains        bench.py --chain full|pixel, 8- and 10-bit
#   txs           tools/bench_txsearch.py, 8- and 10-bit, cdef_dist and transform-domain distortion
#   txs_ab:LIBS   the same with each library of the comma-separated list copied in (same-box A/B)
#   lrf_ab:LIBS   tools/bench_lrf_search.py per library
#   stage_ab:LIBS STAGES=... tools/frame_pipeline.py --stages per library
#   soak          tools/gpu_soak.py (type search x 12 rounds, everything x 3), tools/me_persist_soak.py (90 s)
#   bench_ab:LIBS the headline bench (bench.py --no-extra $BENCH_ARGS) per library
#   px_ab:LIBS    bench.py --chain pixel, 8- and 10-bit, per library
#   dry           bench.py --gpus 2 / 4 --single-device (control flow of the N > 1 path on one GPU)
#   kernels       tools/bench_kernels.py, 8- and 10-bit
#   frame         tools/frame_pipeline.py, 8- and 10-bit
#   me            tools/bench_me.py
#   prof          rocprofv3 --kernel-trace --stats of bench.py (100 steps) and of tools/bench_txsearch.py
#   pmc_txs       SQ counters of the fan-out launches (separate --pmc passes, no trace domains)
#   pmc_hbm       FETCH_SIZE / WRITE_SIZE (separate passes) of the 10-bit frame and of the type search
#   pmc_px        SQ / LDS counters of the pixel chain
#   pmc_frame     SQ / LDS / HBM counters of every kernel of the config-4 frame (tools/frame_pipeline.py)
#   pmc_stage_hbm[:BD]  STAGES=substr: FETCH_SIZE of the kernels of those stages of tools/frame_pipeline.py
#   pmc_lrf[:BD]  SQ / LDS counters of the restoration search alone (tools/bench_lrf_search.py); pmc_lrf_hbm[:BD]: its FETCH_SIZE
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT || exit 1
cp rav1e_amd/librav1e_hip.so /tmp/lib_orig.so

pmc_pass() {   # pmc_pass NAME "COUNTERS" -- cmd...   (one rocprofv3 run per counter group; csv kept small)
  local name=$1 ctrs=$2; shift 2; shift
  (cd /tmp && timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- "$@" > /tmp/pmc_$name.log 2>&1)
  local f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_rows.py "$f" > $OUT/pmc_$name.json || echo "no counters for $name (see /tmp/pmc_$name.log)"; tail -2 /tmp/pmc_$name.log > $OUT/pmc_$name.log
}

for STEP in "$@"; do
  ARG=${STEP#*:}; [ "$ARG" = "$STEP" ] && ARG=""
  case ${STEP%%:*} in
    tests)
      echo "== pytest -m gpu ${ARG:+-k $ARG}"
      if [ -n "$ARG" ]; then timeout 1800 python -m pytest tests -m gpu -x -q -k "$ARG" 2>&1 | tail -15 | tee $OUT/pytest_gpu_k.log
      else timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.log; fi ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log ;;
    bench) timeout 1200 python bench.py --detail-out gpurun_out/$TAG/bench_detail.json 2>$OUT/bench.err | grep "^{" | tee $OUT/bench.json | cut -c1-400
           python3 -c "
import json; s=open('$OUT/bench.json').read(); d=json.loads(s); print('compact bytes', len(s), 'lines', len(s.strip().splitlines()), 'roofline' in d, 'cpu_baseline' in d)
for e in d['extra_lines']: print(' ', e['name'], e['value'], e.get('unit','Mpx/s'), e['ms_per_step'], e['parity_ok'], e.get('cpu'))" ;;
    bench_driver)   # exactly the driver's command
           timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out gpurun_out/$TAG/bench_driver_detail.json 2>$OUT/bench_driver.err | tee $OUT/bench_driver.json | wc -c ;;
    chains)
      for bd in 8 10; do
        [ $bd = 10 ] && timeout 600 python bench.py --cpu-seconds 0 --no-extra --bit-depth 10 --detail-out gpurun_out/$TAG/bench_10bit_detail.json 2>/dev/null | grep "^{" > $OUT/bench_10bit.json
        for chain in full pixel; do
          timeout 600 python bench.py --cpu-seconds 0 --chain $chain --bit-depth $bd --detail-out gpurun_out/$TAG/${chain}_chain_${bd}bit_detail.json 2>/dev/null | grep "^{" > $OUT/${chain}_chain_${bd}bit.json
          python3 -c "
import json; d=json.loads(open('$OUT/${chain}_chain_${bd}bit.json').read()); print('$chain $bd', d.get('value'), d.get('kernel_ms'), d.get('rdo_only',{}).get('value'))"
        done
      done ;;
    txs)
      for bd in 8 10; do for kind in 3 0; do
        timeout 600 python tools/bench_txsearch.py --bit-depth $bd --kind $kind 2>&1 | grep "^{" | tee -a $OUT/txsearch.jsonl | cut -c1-330
      done; done ;;
    txs_ab)   # TXS_SIZES=8 TXS_PASSES=1 narrow the run
      for pass in $(seq 1 ${TXS_PASSES:-2}); do for lib in ${ARG//,/ }; do
        cp $lib rav1e_amd/librav1e_hip.so
        for bd in 8 10; do
          timeout 600 python tools/bench_txsearch.py --bit-depth $bd --kind 3 --sizes ${TXS_SIZES:-8,16,32} 2>&1 | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('%-22s bd %2d %2dx%-2d fan %.4f ind %.4f ratio %.3f ok %s' % ('$lib'.split('/')[-1], d['bd'], d['size'], d['size'], d['fanout_ms'], d['independent_ms'], d['ratio'], d['slots_equal_independent']))"
        done
      done; done 2>&1 | tee $OUT/txs_ab.txt
      cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so ;;
    lrf_ab)
      for pass in 1 2; do for lib in ${ARG//,/ }; do
        cp $lib rav1e_amd/librav1e_hip.so
        for bd in 8 10; do echo "$(basename $lib) $(timeout 300 python tools/bench_lrf_search.py --bit-depth $bd 2>/dev/null | grep '^{')"; done
      done; done 2>&1 | tee $OUT/lrf_ab.txt
      cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so ;;
    stage_ab)   # STAGES=substr,substr [PASSES="1 2"]: tools/frame_pipeline.py --stages per library, 8- and 10-bit
      for pass in ${PASSES:-1 2}; do for lib in ${ARG//,/ }; do
        cp $lib rav1e_amd/librav1e_hip.so
        for bd in 8 10; do echo "$(basename $lib) $(timeout 300 python tools/frame_pipeline.py --bit-depth $bd --stages "$STAGES" 2>/dev/null | grep '^{')"; done
      done; done 2>&1 | tee $OUT/stage_ab.txt
      cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so ;;
    me_grid)   # the persistent tile ME at different grid sizes / spin settings (environment knobs of csrc/me.hip)
      for pass in 1 2; do for g in 2048 1024 1536 3072 4096; do for bd in 8 10; do
        echo "grid $g $(R1_ME_PERSISTENT_GRID=$g timeout 300 python tools/frame_pipeline.py --bit-depth $bd --stages estimate_tile 2>/dev/null | grep '^{')"
      done; done; done 2>&1 | tee $OUT/me_grid.txt ;;
    soak)   # seeded GPU parity tests re-run with shifted seeds (type search, chains, ME), then the persistent ME against the diagonal launches
      timeout 400 python tools/gpu_soak.py --rounds ${SOAK_ROUNDS:-12} -k txsearch 2>&1 | tail -3 | tee $OUT/soak_txsearch.log
      timeout 400 python tools/gpu_soak.py --rounds 3 2>&1 | tail -3 | tee $OUT/soak_all.log
      timeout 200 python tools/me_persist_soak.py --seconds 90 2>&1 | tail -3 | tee $OUT/soak_me_persist.log
      timeout 200 python tools/loop_decision_soak.py --seconds 90 2>&1 | tail -2 | tee $OUT/soak_loop_decision.log ;;
    bench_ab)   # the headline bench per library (BENCH_ARGS, e.g. "--bit-depth 10"), two passes
      for pass in 1 2; do for lib in ${ARG//,/ }; do
        cp $lib rav1e_amd/librav1e_hip.so
        timeout 600 python bench.py --no-extra --cpu-seconds 0 $BENCH_ARGS 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-22s %9.0f Mpx/s kernel_ms %s ok %s' % ('$lib'.split('/')[-1], d['value'], d['kernel_ms'], d.get('parity_ok')))"
      done; done 2>&1 | tee $OUT/bench_ab.txt
      cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so ;;
    px_ab)
      for pass in 1 2; do for lib in ${ARG//,/ }; do
        cp $lib rav1e_amd/librav1e_hip.so
        for bd in 8 10; do
          timeout 600 python bench.py --no-extra --cpu-seconds 0 --bit-depth $bd --chain pixel 2>/dev/null | grep '^{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-22s bd %2d %9.0f Mpx/s rdo_only %s kernel_ms %s ok %s' % ('$lib'.split('/')[-1], $bd, d['value'], d.get('rdo_only',{}).get('value'), d['kernel_ms'], d.get('parity_ok')))"
        done
      done; done 2>&1 | tee $OUT/px_ab.txt
      cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so ;;
    dry)   # N = 2 and 4 control flow on ONE GPU (numbers mean nothing): the exchange, its pre-run and after-run self-checks
      for n in 2 4 2o; do
        extra=""; [ $n = 2o ] && extra="--overlap-exchange"
        timeout 600 python bench.py --gpus ${n%o} --single-device --backend gloo --steps 30 --warmup 3 --cpu-seconds 0 --no-extra $extra --detail-out gpurun_out/$TAG/dry${n}_detail.json 2>$OUT/dry$n.err | grep "^{" > $OUT/dry$n.json
        python3 -c "
import json; d=json.loads(open('$OUT/dry$n.json').read()); print('dry $n', d['value'], d['config']['exchange'], d['config']['exchange_ok'])" || tail -5 $OUT/dry$n.err
      done ;;
    kernels) for bd in 8 10; do timeout 900 python tools/bench_kernels.py --bit-depth $bd 2>/dev/null | grep "^{" > $OUT/kernels_${bd}bit.jsonl; wc -l $OUT/kernels_${bd}bit.jsonl; done ;;
    frame) for bd in 8 10; do timeout 900 python tools/frame_pipeline.py --verify --bit-depth $bd 2>$OUT/frame_${bd}.err | grep "^{" | tee $OUT/frame_pipeline_${bd}bit.json | cut -c1-600; done ;;
    me) timeout 900 python tools/bench_me.py --cpu 2>/dev/null | grep "^{" | tee $OUT/me_4k.jsonl | cut -c1-200 ;;
    prof)
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o prof -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --cpu-seconds 0 --no-extra > /tmp/prof_$TAG.log 2>&1; tail -1 /tmp/prof_$TAG.log | cut -c1-200)
      find /tmp/prof_$TAG -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \; 2>/dev/null
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft_$TAG -o prof -- python $GRAFT_REPO_ROOT/tools/bench_txsearch.py --bit-depth 10 --kind 3 > /tmp/proft_$TAG.log 2>&1; tail -3 /tmp/proft_$TAG.log | cut -c1-200)
      find /tmp/proft_$TAG -name "*kernel_stats*" -exec cp {} $OUT/txsearch_kernel_stats.csv \; 2>/dev/null ;;
    pmc_txs)
      for bd in 8 10; do
        pmc_pass txs${bd}_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" -- python $GRAFT_REPO_ROOT/tools/bench_txsearch.py --bit-depth $bd --kind 3 --fanout-only --reps 3 --sustain-ms 0
        pmc_pass txs${bd}_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" -- python $GRAFT_REPO_ROOT/tools/bench_txsearch.py --bit-depth $bd --kind 3 --fanout-only --reps 3 --sustain-ms 0
      done ;;
    pmc_hbm)   # FETCH_SIZE and WRITE_SIZE in SEPARATE passes (together they do not fit one pass and the run hangs until its timeout)
      FP="python $GRAFT_REPO_ROOT/tools/frame_pipeline.py --bit-depth 10 --reps 2 --sustain-ms 0"
      TX="python $GRAFT_REPO_ROOT/tools/bench_txsearch.py --bit-depth 10 --kind 3 --fanout-only --reps 3 --sustain-ms 0"
      PMC_TIMEOUT=150 pmc_pass frame10_fetch "FETCH_SIZE" -- $FP
      PMC_TIMEOUT=150 pmc_pass frame10_write "WRITE_SIZE" -- $FP
      PMC_TIMEOUT=120 pmc_pass txs10_fetch "FETCH_SIZE" -- $TX
      PMC_TIMEOUT=120 pmc_pass txs10_write "WRITE_SIZE" -- $TX ;;
    pmc_px)
      for bd in 8 10; do
        pmc_pass px${bd}_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --prewarm-ms 0 --cpu-seconds 0 --no-extra --chain pixel --bit-depth $bd
        pmc_pass px${bd}_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --prewarm-ms 0 --cpu-seconds 0 --no-extra --chain pixel --bit-depth $bd
      done ;;
    pmc_frame)   # every kernel of the config-4 frame (ME, pre-screen, chains, type search, deblock, CDEF, LRF search / filter)
      for bd in 10 8; do
        FP="python $GRAFT_REPO_ROOT/tools/frame_pipeline.py --bit-depth $bd --reps 2 --sustain-ms 0"
        pmc_pass frame${bd}_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" -- $FP
        pmc_pass frame${bd}_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" -- $FP
        [ $bd = 10 ] && pmc_pass frame${bd}_d "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM" -- $FP
      done ;;
    pmc_lrf)     # the restoration search alone (tools/bench_lrf_search.py), 8-bit then 10-bit
      for bd in ${ARG:-8}; do
        LB="python $GRAFT_REPO_ROOT/tools/bench_lrf_search.py --bit-depth $bd --reps 3 --sustain-ms 0"
        PMC_TIMEOUT=120 pmc_pass lrf${bd}_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" -- $LB
        PMC_TIMEOUT=120 pmc_pass lrf${bd}_b "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" -- $LB
      done ;;
    pmc_lrf_hbm)   # FETCH_SIZE of the restoration search alone (its own pass, as the guide prescribes)
      for bd in ${ARG:-10}; do
        LB="python $GRAFT_REPO_ROOT/tools/bench_lrf_search.py --bit-depth $bd --reps 3 --sustain-ms 0"
        PMC_TIMEOUT=120 pmc_pass lrf${bd}_fetch "FETCH_SIZE" -- $LB
      done ;;
    pmc_stage_hbm)   # STAGES=substr: FETCH_SIZE of the kernels of those frame stages, 10-bit (its own pass)
      FP="python $GRAFT_REPO_ROOT/tools/frame_pipeline.py --bit-depth ${ARG:-10} --reps 2 --sustain-ms 0 --stages $STAGES"
      PMC_TIMEOUT=60 pmc_pass stage_fetch "FETCH_SIZE" -- $FP ;;
    me_step)   # ARG = a library built with -DR1_ME_PROF: where a block search of the persistent tile ME spends its time
      cp $ARG rav1e_amd/librav1e_hip.so
      for bd in 8 10; do R1_ME_PROF_BD=$bd timeout 300 python tools/me_prof.py 4 2>/dev/null | grep "^{" | sed "s/^/bd $bd /"; done | tee $OUT/me_step.txt
      cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so ;;
    prof_frame)   # rocprofv3 --kernel-trace --stats of the config-4 frame (tools/frame_pipeline.py, 10-bit) -> frame_kernel_stats.csv
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proff_$TAG -o prof -- python $GRAFT_REPO_ROOT/tools/frame_pipeline.py --bit-depth ${ARG:-10} --reps 20 > /tmp/proff_$TAG.log 2>&1; tail -1 /tmp/proff_$TAG.log | cut -c1-200)
      find /tmp/proff_$TAG -name "*kernel_stats*" -exec cp {} $OUT/frame_kernel_stats.csv \; 2>/dev/null
      head -30 $OUT/frame_kernel_stats.csv | cut -c1-160 ;;
    pmc_lines)   # FETCH_SIZE / WRITE_SIZE / VALU counters of EVERY launch of bench.py, its extra / config / frame lines
                 # included (three passes; csv rows of our kernels only) -> pmc_launches.json (bench.py launch_pmc reads
                 # the newest profiles/r*_pmc_launches.json)
      mkdir -p $OUT/pmcl
      for pass in "tcc1 FETCH_SIZE" "tcc2 WRITE_SIZE" "sq1 SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
        set -- $pass; nm=$1; shift
        (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $@ --output-format csv -d /tmp/pmcl_$TAG/$nm -o p -- \
          python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --prewarm-ms 0 --cpu-seconds 0 --no-events > /tmp/pmcl_$TAG.$nm.log 2>&1)
        f=$(find /tmp/pmcl_$TAG/$nm -name "*counter_collection.csv" | head -1)
        [ -n "$f" ] && (head -1 $f; grep -E 'k_[a-z]' $f) > $OUT/pmcl/$nm.csv || (echo "no counters for $nm"; tail -5 /tmp/pmcl_$TAG.$nm.log)
      done
      python3 tools/pmc_summary.py $OUT/pmcl $OUT/pmc_lines_summary.json $OUT/pmc_launches.json | tail -3
      rm -rf $OUT/pmcl ;;
    *) echo "unknown step $STEP" ;;
  esac
done
cp /tmp/lib_orig.so rav1e_amd/librav1e_hip.so
ls $OUT
