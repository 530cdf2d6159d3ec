#!/bin/bash
# One gpurun call.  Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh <tag> [stages...]'
# stages: smoke tests probe bench pmc trace   (default: all)
TAG=${1:-r01}; shift
STAGES=${@:-smoke tests probe bench trace pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$PWD
has() { [[ " $STAGES " == *" $1 "* ]]; }
echo "== host: $(nproc) cpus" ; rocminfo 2>/dev/null | grep -m1 -E "gfx9" 
if has smoke; then echo "== smoke"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 $OUT/smoke.log; fi
if has quick; then echo "== pytest quick: ${QUICK:-tests/test_gpu_adapt.py}"
  timeout ${QUICK_TIMEOUT:-300} python -m pytest ${QUICK:-tests/test_gpu_adapt.py} -m gpu -q --durations=12 > $OUT/pytest_quick.log 2>&1 ; echo "pytest rc=$?" ; tail -30 $OUT/pytest_quick.log | cut -c1-250; fi
if has tests; then echo "== pytest -m gpu"
  timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -25 $OUT/pytest_gpu.log | cut -c1-300; fi
if has probe; then echo "== kernel probes"
  timeout 300 python scripts/kernel_probe.py adv --size 512 --variants ${ADV_VARIANTS:-0,2,3} > $OUT/probe_adv_512.jsonl 2>&1 ; cat $OUT/probe_adv_512.jsonl
  timeout 300 python scripts/kernel_probe.py adv --size 256 --variants 0 > $OUT/probe_adv_256.jsonl 2>&1 ; cat $OUT/probe_adv_256.jsonl
  timeout 300 python scripts/kernel_probe.py pre --size 256 > $OUT/probe_pre_256.jsonl 2>&1 ; cat $OUT/probe_pre_256.jsonl; fi
if has benchq; then echo "== bench 512 (no cpu baseline)"
  timeout 300 python bench.py --size 512 --steps 5 --warmup 1 --no-cpu --stencil-only > $OUT/benchq_512_stencil.json 2> $OUT/benchq_512_stencil.err ; tail -c 1500 $OUT/benchq_512_stencil.json
  timeout 600 python bench.py --no-cpu > $OUT/benchq_512.json 2> $OUT/benchq_512.err ; echo "bench rc=$?" ; tail -c 3000 $OUT/benchq_512.json ; tail -3 $OUT/benchq_512.err
  fi
if has amr; then echo "== bench --amr (multi-level mesh built on the device)"
  timeout 600 python bench.py --amr --steps 5 --warmup 1 > $OUT/bench_amr.json 2> $OUT/bench_amr.err ; echo "bench rc=$?" ; tail -c 3500 $OUT/bench_amr.json ; tail -3 $OUT/bench_amr.err
  timeout 600 python bench.py --amr --steps 5 --warmup 1 --block-solver 1 > $OUT/bench_amr_fdm.json 2> $OUT/bench_amr_fdm.err ; tail -c 600 $OUT/bench_amr_fdm.json | head -c 600
  fi
if has bench; then echo "== bench 512 stencil-only"
  timeout 300 python bench.py --size 512 --steps 5 --warmup 1 --no-cpu --stencil-only > $OUT/bench_512_stencil.json 2> $OUT/bench_512_stencil.err ; tail -c 1200 $OUT/bench_512_stencil.json
  echo "== bench default (512 full + cpu baseline)"
  timeout 1200 python bench.py > $OUT/bench_512.json 2> $OUT/bench_512.err ; echo "bench rc=$?" ; tail -c 4000 $OUT/bench_512.json ; tail -3 $OUT/bench_512.err
  fi
if has trace; then echo "== rocprofv3 kernel trace of the default bench command (512^3 full step, no cpu baseline)"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -o t -- python $ROOT/bench.py --no-cpu --no-alt > $ROOT/$OUT/trace.log 2>&1 )
  for f in $(find $OUT/trace -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats_512_full.csv; head -14 $f | cut -c1-200; done
  grep -E "^\{" $OUT/trace.log | tail -c 1500
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace512 -o t -- python $ROOT/scripts/kernel_probe.py one --size 512 --kernel adv > $ROOT/$OUT/trace512.log 2>&1 )
  for f in $(find $OUT/trace512 -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats_512_adv.csv; head -6 $f | cut -c1-200; done
  rm -rf $OUT/trace/*/*.db $OUT/trace512/*/*.db 2>/dev/null; fi
if has pmc; then echo "== rocprofv3 PMC passes (advdiff 512^3): FETCH_SIZE, WRITE_SIZE in separate runs"
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$C -o p -- python $ROOT/scripts/kernel_probe.py one --size 512 --kernel adv > $ROOT/$OUT/pmc_$C.log 2>&1 )
    for f in $(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1); do python - "$f" $C <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") == c:
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(c, k, "launches", len(v), "mean", sum(v) / len(v))
PY
    done
  done; fi
if has pmc2; then echo "== rocprofv3 PMC passes (one pressure projection at 512^3): FETCH_SIZE, WRITE_SIZE in separate runs"
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/$OUT/pmc2_$C -o p -- python $ROOT/scripts/kernel_probe.py one --size 512 --kernel solve > $ROOT/$OUT/pmc2_$C.log 2>&1 )
    for f in $(find $OUT/pmc2_$C -name "*counter_collection.csv" | head -1); do python - "$f" $C <<'PY' | tee -a $OUT/pmc2_summary.txt
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") == c:
        acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(c, k, "launches", len(v), "mean_KB", round(sum(v) / len(v), 1))
PY
    done
    rm -rf $OUT/pmc2_$C
  done; fi
if has pmc3; then echo "== rocprofv3 SQ counters of the block preconditioners (256^3, solver-like input)"
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE"; do
    TAGC=$(echo $SET | cut -d" " -f1)
    ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $ROOT/$OUT/pmc3_$TAGC -o p -- python $ROOT/scripts/kernel_probe.py pre --size 256 --reps 2 > $ROOT/$OUT/pmc3_$TAGC.log 2>&1 )
    for f in $(find $OUT/pmc3_$TAGC -name "*counter_collection.csv" | head -1); do python - "$f" <<'PY' | tee -a $OUT/pmc3_summary.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:48]
    if "k_precond" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    for c, v in sorted(d.items()):
        print(k, c, "launches", len(v), "mean", round(sum(v) / len(v), 1), "last", v[-1])
PY
    done
    rm -rf $OUT/pmc3_$TAGC
  done; fi
echo "== done"
