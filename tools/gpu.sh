#!/bin/bash
# ONE parameterised lease script (replaces the per-call tools/gpu_r0N_*.sh one-offs of rounds 1-3).
#   gpurun --timeout 1500 -- 'bash tools/gpu.sh TAG step [step ...]'
# Steps run in the order given; every artefact lands in gpurun_out/ with TAG in its name.
#   test[:EXPR]        pytest -m gpu (optionally -k EXPR; spaces in EXPR as '+')
#   testfile:PATH      pytest PATH -m gpu
#   smoke              __graft_entry__.smoke()
#   bench[:ARGS]       the driver's bench command (ARGS: extra flags, '+' for spaces) -> bench_TAG.json
#   only:POINT[:K]     bench.py --only POINT --steps K (default 4), JSON -> bench_TAG_POINT.json
#   prof:POINT[:K]     rocprofv3 --kernel-trace over `bench.py --only POINT` -> TAG_POINT_kernel_stats.csv (+ step sequence)
#   trace:POINT        the same with the HIP/RCCL stream timeline kept: TAG_POINT_overlap.csv (kernel start/end per stream)
#   pmc:POINT          three PMC passes (FETCH_SIZE | WRITE_SIZE | MFMA-busy + clock) -> TAG_POINT_pmc_tables.md
#   py:SCRIPT[:ARGS]   python SCRIPT ARGS > py_TAG_<basename>.log
#   pyprof:SCRIPT[:ARGS]  the same under rocprofv3 --kernel-trace -> TAG_<basename>_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
TAG=$1; shift
SECONDS=0
cd $R
unplus() { echo "$1" | tr '+' ' '; }
brief() {   # one-screen summary of a bench JSON line
python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as ex:
    print("no JSON:", ex); sys.exit(0)
keep = ("value", "ms_per_step", "peak_vram_gb", "tokens_per_s", "ms_per_token", "gemm_frac_of_mfma_peak", "attention_frac_of_mfma_peak",
        "attention_fwd_frac_of_mfma_peak", "attention_bwd_frac_of_mfma_peak", "attention_share_of_step", "frac_of_hbm_peak", "error")
print({k: d[k] for k in ("value", "ms_per_step", "peak_vram_gb", "steps", "loss_first_last") if k in d},
      {k: v for k, v in d.items() if k in keep[5:]})
if d.get("roofline"):
    r = d["roofline"]; print(" roofline", r["achieved"], r["frac"], r.get("attention"))
if d.get("config"):
    print(" gc", d["config"].get("gradient_checkpointing"), d["config"].get("gc_schedule_chosen"))
for k, v in (d.get("alt") or {}).items():
    print("  ", k[:58].ljust(58), {kk: vv for kk, vv in v.items() if kk in keep})
PY
}
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  case $kind in
    test)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$(unplus "$arg")" > $OUT/pytest_$TAG.log 2>&1
      else timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_$TAG.log 2>&1; fi
      echo "[$step] rc=$? ($SECONDS s)"; tail -6 $OUT/pytest_$TAG.log ;;
    testfile)
      n=$(basename ${arg%%:*} .py)
      timeout 1200 python -m pytest $(unplus "$arg") -m gpu -q -x --durations=6 > $OUT/pytest_${TAG}_$n.log 2>&1
      echo "[$step] rc=$? ($SECONDS s)"; tail -12 $OUT/pytest_${TAG}_$n.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ;;
    bench)
      timeout 1200 python bench.py $(unplus "$arg") > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
      echo "[$step] rc=$? ($SECONDS s)"; brief $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err ;;
    only)
      P=${arg%%:*}; K=4; [ "$P" != "$arg" ] && K=${arg#*:}
      timeout 900 python bench.py --only $P --steps $K --warmup 3 > $OUT/bench_${TAG}_$P.json 2> $OUT/bench_${TAG}_$P.err
      echo "[$step] rc=$? ($SECONDS s)"; brief $OUT/bench_${TAG}_$P.json; tail -2 $OUT/bench_${TAG}_$P.err ;;
    prof|trace)
      P=${arg%%:*}; K=3; [ "$P" != "$arg" ] && K=${arg#*:}
      cd /tmp
      EXTRA=""; [ $kind = trace ] && EXTRA="--rccl-trace --hip-runtime-trace"
      timeout 900 rocprofv3 --kernel-trace $EXTRA -d $OUT/prof_${TAG}_$P -o bench -- python $R/bench.py --only $P --steps $K --warmup 1 \
          --no-cpu-baseline --no-gpu-baseline --alt-steps 0 > $OUT/prof_${TAG}_$P.log 2>&1
      echo "[$step] rc=$? ($SECONDS s)"
      cd $R
      DB=$(find $OUT/prof_${TAG}_$P -name '*.db' | head -1)
      if [ -n "$DB" ]; then
        python tools/rocpd_stats.py $DB > $OUT/${TAG}_${P}_kernel_stats.csv 2> $OUT/${TAG}_${P}_stats.err
        head -14 $OUT/${TAG}_${P}_kernel_stats.csv | cut -c1-150
        python tools/rocpd_sequence.py $DB > $OUT/${TAG}_${P}_step_sequence.csv 2> $OUT/${TAG}_${P}_seq.err
        grep "^# kernels" $OUT/${TAG}_${P}_step_sequence.csv
        if [ $kind = trace ]; then
          python tools/rocpd_overlap.py $DB > $OUT/${TAG}_${P}_overlap.md 2> $OUT/${TAG}_${P}_overlap.err
          head -40 $OUT/${TAG}_${P}_overlap.md | cut -c1-170
        fi
      else tail -5 $OUT/prof_${TAG}_$P.log; fi
      rm -rf $OUT/prof_${TAG}_$P ;;
    pmc)
      P=$arg
      cd /tmp
      BENCH="python $R/bench.py --only $P --steps 2 --warmup 1 --alt-steps 0 --no-cpu-baseline --no-gpu-baseline"
      for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
        set -- $pass; t=$1; shift
        timeout 420 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$t -o pmc -- $BENCH > $OUT/pmc_${TAG}_$t.log 2>&1
        echo "[pmc $t] rc=$? ($SECONDS s)"
        DB=$(find $OUT/pmc_$t -name '*.db' | head -1)
        [ -n "$DB" ] && python $R/tools/pmc_summary.py $DB > $OUT/${TAG}_${P}_pmc_$t.txt 2> $OUT/pmc_$t.err
        rm -rf $OUT/pmc_$t
      done
      cd $R
      [ "$P" = primary ] && python tools/pmc_to_json.py $OUT/${TAG}_${P}_pmc_fetch.txt $OUT/${TAG}_${P}_pmc_write.txt > $OUT/pmc_traffic.json 2> $OUT/pmc_json.err
      {
        echo "# PMC tables, \`bench.py --only $P --steps 2 --warmup 1\` ($TAG; tools/gpu.sh pmc:$P)"
        echo; echo "## MFMA-busy and effective clock"; echo
        python tools/pmc_tables.py mfma $OUT/${TAG}_${P}_pmc_mfma.txt
        echo; echo "## HBM-side traffic per launch"; echo
        python tools/pmc_tables.py hbm $OUT/${TAG}_${P}_pmc_fetch.txt $OUT/${TAG}_${P}_pmc_write.txt 8192
      } > $OUT/${TAG}_${P}_pmc_tables.md 2> $OUT/pmc_tables.err
      head -24 $OUT/${TAG}_${P}_pmc_tables.md | cut -c1-170 ;;
    py)
      S=${arg%%:*}; A=""; [ "$S" != "$arg" ] && A=$(unplus "${arg#*:}")
      timeout 1200 python $S $A > $OUT/py_${TAG}_$(basename $S .py).log 2>&1
      echo "[$step] rc=$? ($SECONDS s)"; tail -25 $OUT/py_${TAG}_$(basename $S .py).log | cut -c1-200 ;;
    pyprof)      # rocprofv3 --kernel-trace over a python script -> TAG_<basename>_kernel_stats.csv
      S=${arg%%:*}; A=""; [ "$S" != "$arg" ] && A=$(unplus "${arg#*:}")
      n=$(basename $S .py)
      cd /tmp
      timeout 900 rocprofv3 --kernel-trace -d $OUT/prof_${TAG}_$n -o prof -- python $R/$S $A > $OUT/pyprof_${TAG}_$n.log 2>&1
      echo "[$step] rc=$? ($SECONDS s)"
      cd $R
      DB=$(find $OUT/prof_${TAG}_$n -name '*.db' | head -1)
      if [ -n "$DB" ]; then
        python tools/rocpd_stats.py $DB > $OUT/${TAG}_${n}_kernel_stats.csv 2> $OUT/${TAG}_${n}_stats.err
        head -16 $OUT/${TAG}_${n}_kernel_stats.csv | cut -c1-150
        python tools/rocpd_tail.py $DB 70 40 > $OUT/${TAG}_${n}_tail.csv 2>> $OUT/${TAG}_${n}_stats.err
      else tail -5 $OUT/pyprof_${TAG}_$n.log; fi
      rm -rf $OUT/prof_${TAG}_$n ;;
    *) echo "unknown step $step" ;;
  esac
done
echo "all done ($SECONDS s)"
