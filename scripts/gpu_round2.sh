#!/bin/bash
# One gpurun call of round 2.  Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round2.sh <tag> [stages...]'
# stages: smoke tests multirank probe benchq bench trace pmc campaign256 campaign512 cpuprobe
TAG=${1:-r02a}; shift
STAGES=${@:-smoke tests multirank probe benchq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$PWD
has() { [[ " $STAGES " == *" $1 "* ]]; }
echo "== host: $(nproc) cpus, $(free -g | awk '/Mem:/{print $2}') GB" ; rocminfo 2>/dev/null | grep -m1 -E "gfx9"
if has smoke; then echo "== smoke"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 $OUT/smoke.log; fi
if has quick; then echo "== pytest quick: ${QUICK:-tests/test_gpu_parity.py}"
  timeout ${QUICK_TIMEOUT:-600} python -m pytest ${QUICK:-tests/test_gpu_parity.py} -m gpu -q -x --durations=8 ${QUICK_ARGS} > $OUT/pytest_quick.log 2>&1 ; echo "pytest rc=$?" ; tail -40 $OUT/pytest_quick.log | cut -c1-300; fi
if has tests; then echo "== pytest -m gpu (all but the thread-per-rank file)"
  timeout 1200 python -m pytest tests -m gpu -q --durations=10 --ignore=tests/test_gpu_multirank.py -s > $OUT/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; grep -E "256\^3|passed|failed|FAILED|Error" $OUT/pytest_gpu.log | tail -40 | cut -c1-300; tail -15 $OUT/pytest_gpu.log | cut -c1-300; fi
if has multirank; then echo "== pytest multirank (thread-per-rank virtual communicator)"
  timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q ${MR_ARGS:--x} --durations=5 > $OUT/pytest_multirank.log 2>&1 ; echo "pytest rc=$?" ; tail -40 $OUT/pytest_multirank.log | cut -c1-300; fi
if has probe; then echo "== kernel probes"
  timeout 300 python scripts/kernel_probe.py pre --size 256 > $OUT/probe_pre_256.jsonl 2>&1 ; cat $OUT/probe_pre_256.jsonl
  timeout 300 python scripts/kernel_probe.py adv --size 512 --variants 0 > $OUT/probe_adv_512.jsonl 2>&1 ; cat $OUT/probe_adv_512.jsonl; fi
if has benchq; then echo "== bench 512 (no cpu baseline)"
  timeout 600 python bench.py --no-cpu --steps ${BENCH_STEPS:-3} --warmup 1 > $OUT/benchq_512.json 2> $OUT/benchq_512.err ; echo "bench rc=$?" ; tail -c 4500 $OUT/benchq_512.json ; tail -3 $OUT/benchq_512.err
  fi
if has cpuprobe; then echo "== reference on the host: threads scan at 256^3 (1 step) for the cpu_baseline setting"
  for T in 32 64 $(nproc); do
    timeout 600 python - $T <<'PY' 2>&1 | tail -2
import sys, os, numpy as np
sys.path.insert(0, "tests")
import oracle_lib as O
T = int(sys.argv[1])
args = O.ref_args((1, 1, 1), 6, 5, 2 * np.pi, ("wall",) * 3, nu=0.01, cfl=0.3, extra=["-rampup", "0"])
import time
t0 = time.time()
recs, _ = O.run_ref(["zero chi", "set step 21", "rep 1", "op steps 1"], args, threads=T, timeout=500)
r = [q for q in recs if q["op"] == "steps"][0]
print("threads", T, "256^3 one step seconds", r["seconds"], "iters", r["iters"], "wall incl. startup", round(time.time() - t0, 1))
PY
  done; fi
if has bench; then echo "== bench default (512 full + cpu baselines)"
  timeout 1500 python bench.py > $OUT/bench_512.json 2> $OUT/bench_512.err ; echo "bench rc=$?" ; tail -c 6000 $OUT/bench_512.json ; tail -3 $OUT/bench_512.err
  echo "== bench 512 stencil-only"
  timeout 300 python bench.py --size 512 --steps 5 --warmup 1 --no-cpu --stencil-only > $OUT/bench_512_stencil.json 2> $OUT/bench_512_stencil.err ; tail -c 1500 $OUT/bench_512_stencil.json
  fi
if has amr; then echo "== bench --amr"
  timeout 600 python bench.py --amr --steps 5 --warmup 1 > $OUT/bench_amr.json 2> $OUT/bench_amr.err ; echo "bench rc=$?" ; tail -c 3500 $OUT/bench_amr.json ; tail -3 $OUT/bench_amr.err; fi
if has trace; then echo "== rocprofv3 kernel trace of the default bench command (512^3 full step, no cpu baseline)"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -o t -- python $ROOT/bench.py --no-cpu --no-alt > $ROOT/$OUT/trace.log 2>&1 )
  for f in $(find $OUT/trace -name "*kernel_stats.csv" | head -1); do cp $f $OUT/kernel_stats_512_full.csv; head -16 $f | cut -c1-200; done
  grep -E "^\{" $OUT/trace.log | tail -c 1500
  rm -rf $OUT/trace/*/*.db 2>/dev/null; fi
if has pmc; then echo "== rocprofv3 PMC passes (one pressure projection at 512^3): FETCH_SIZE, WRITE_SIZE in separate runs"
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/$OUT/pmc2_$C -o p -- python $ROOT/scripts/kernel_probe.py one --size 512 --kernel solve > $ROOT/$OUT/pmc2_$C.log 2>&1 )
    for f in $(find $OUT/pmc2_$C -name "*counter_collection.csv" | head -1); do python - "$f" $C <<'PY' | tee -a $OUT/pmc_solver_summary.txt
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
if has pmcsq; then echo "== rocprofv3 SQ counters of the block preconditioners (256^3, solver-like input)"
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
    TAGC=$(echo $SET | cut -d" " -f1)
    ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $ROOT/$OUT/pmc3_$TAGC -o p -- python $ROOT/scripts/kernel_probe.py pre --size 256 --reps 2 > $ROOT/$OUT/pmc3_$TAGC.log 2>&1 )
    for f in $(find $OUT/pmc3_$TAGC -name "*counter_collection.csv" | head -1); do python - "$f" <<'PY' | tee -a $OUT/pmc_block_cg_sq.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    if "k_precond" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    for c, v in sorted(d.items()):
        print(k, c, "launches", len(v), "mean", round(sum(v) / len(v), 1), "last", v[-1])
PY
    done
    tail -5 $OUT/pmc3_$TAGC.log
    rm -rf $OUT/pmc3_$TAGC
  done; fi
if has pmcsq2; then echo "== rocprofv3 SQ counters of the fused solver kernels (one pressure projection at 256^3)"
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    TAGC=$(echo $SET | cut -d" " -f1)
    ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $ROOT/$OUT/pmc4_$TAGC -o p -- python $ROOT/scripts/kernel_probe.py one --size 256 --kernel solve > $ROOT/$OUT/pmc4_$TAGC.log 2>&1 )
    for f in $(find $OUT/pmc4_$TAGC -name "*counter_collection.csv" | head -1); do python - "$f" <<'PY' | tee -a $OUT/pmc_fused_kernels_sq.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:48]
    if "k_loop1_cg" in k or "k_loop2_cg" in k or "k_precond<" in k or "k_lhs" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    for c, v in sorted(d.items()):
        print(k, c, "launches", len(v), "mean", round(sum(v) / len(v), 1))
PY
    done
    tail -3 $OUT/pmc4_$TAGC.log
    rm -rf $OUT/pmc4_$TAGC
  done; fi
if has campaign256; then echo "== campaign: 256^3 periodic, 3 default-tolerance steps + tight projection, device vs compiled reference"
  timeout 1500 python scripts/campaigns/baseline_sizes_vs_reference.py --size 256 --bc periodic --steps 3 --tight --threads ${REF_THREADS:-64} --out profiles/r02/reference_steps_256_periodic.json > $OUT/campaign256.json 2> $OUT/campaign256.err ; echo "rc=$?"; tail -c 3000 $OUT/campaign256.json; tail -3 $OUT/campaign256.err
  cp profiles/r02/reference_steps_256_periodic.json $OUT/ 2>/dev/null; fi
if has campaign512; then echo "== campaign: 512^3 all-wall, ONE reference step, device vs compiled reference"
  timeout 2400 python scripts/campaigns/baseline_sizes_vs_reference.py --size 512 --bc wall --threads ${REF_THREADS:-64} --out profiles/r02/reference_step_512.json > $OUT/campaign512.json 2> $OUT/campaign512.err ; echo "rc=$?"; tail -c 2000 $OUT/campaign512.json; tail -3 $OUT/campaign512.err
  cp profiles/r02/reference_step_512.json $OUT/ 2>/dev/null; fi
echo "== done"
