#!/bin/bash
# One gpurun call of round 3.  Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round3.sh <tag> [stages...]'
# stages: smoke newtests tests multirank bench128 bench256 bench512 flavours window256 trace pmc probe
TAG=${1:-r03a}; shift
STAGES=${@:-smoke newtests bench128 bench256}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
has() { [[ " $STAGES " == *" $1 "* ]]; }
echo "== host: $(nproc) cpus, $(free -g | awk '/Mem:/{print $2}') GB, devices: $(python -c 'import cup3d_amd.capi as c; print(c.device_count())' 2>/dev/null)"
if has smoke; then echo "== smoke (release build)"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 $OUT/smoke.log; fi
if has newtests; then echo "== pytest: round-3 tests first"
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_release_flavour.py tests/test_gpu_rccl.py -m gpu -q --durations=8 -s ${NEW_ARGS} > $OUT/pytest_new.log 2>&1 ; echo "pytest rc=$?"
  grep -E "level [0-9]|passed|failed|FAILED|Error|assert" $OUT/pytest_new.log | tail -60 | cut -c1-400; fi
if has tests; then echo "== pytest -m gpu (everything)"
  timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; grep -E "passed|failed|FAILED|Error" $OUT/pytest_gpu.log | tail -40 | cut -c1-300; tail -12 $OUT/pytest_gpu.log | cut -c1-300; fi
if has resttests; then echo "== pytest -m gpu (all but the files of newtests)"
  timeout 1500 python -m pytest tests -m gpu -q --durations=10 --ignore=tests/test_gpu_parity.py --ignore=tests/test_gpu_release_flavour.py --ignore=tests/test_gpu_rccl.py > $OUT/pytest_rest.log 2>&1 ; echo "pytest rc=$?" ; grep -E "passed|failed|FAILED|Error" $OUT/pytest_rest.log | tail -40 | cut -c1-300; tail -12 $OUT/pytest_rest.log | cut -c1-300; fi
summ() { python - "$1" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("  (no JSON)", e); sys.exit(0)
c = r["config"]
its = c.get("bicgstab_iters_per_step") or 0
print("  value", r["value"], "ms/step", r["ms_per_step"], "its/step", its, "ms/iteration", round(r["ms_per_step"] / its, 4) if its else None, "lib", c.get("library"))
print("  checksum", (c.get("checksum") or {}).get("ok"), "comm", c.get("communication"))
for k in r.get("kernels", [])[:7]:
    print("   ", k["kernel"], k["launches"], k["avg_ms"], k.get("frac"))
PY
}
for S in 128 256 512; do
  if has bench$S; then echo "== bench $S (release build, no cpu baseline, no alt)"
    timeout 900 python bench.py --size $S --no-cpu --no-alt --no-pcie --steps ${BENCH_STEPS:-10} --warmup ${BENCH_WARMUP:-3} > $OUT/bench_$S.json 2> $OUT/bench_$S.err ; echo "bench rc=$?" ; summ $OUT/bench_$S.json ; tail -2 $OUT/bench_$S.err
  fi
done
if has flavours; then echo "== release vs testing build, 256^3, no per-kernel events"
  for F in release testing release testing; do
    CUP3D_HIP_FLAVOUR=$F timeout 600 python bench.py --size 256 --no-cpu --no-alt --no-pcie --no-profile --no-checksum --steps 10 --warmup 3 > $OUT/flavour_$F.json 2>> $OUT/flavour.err ; echo "$F rc=$?"; summ $OUT/flavour_$F.json; done; fi
if has ablhs; then echo "== A/B: LHS as k_lhs launches (no_fuse_lhs=1, testing build) vs inside the loop kernels (default), same box"
  for S in ${AB_SIZES:-256 512}; do
    timeout 900 python bench.py --size $S --no-cpu --no-alt --no-pcie --no-checksum --debug-option no_fuse_lhs=1 --steps ${BENCH_STEPS:-10} --warmup ${BENCH_WARMUP:-3} > $OUT/ab_${S}_klhs.json 2>> $OUT/ab.err; echo "k_lhs launches, $S:"; summ $OUT/ab_${S}_klhs.json
    timeout 900 python bench.py --size $S --no-cpu --no-alt --no-pcie --no-checksum --debug-option no_fuse_lhs=0 --steps ${BENCH_STEPS:-10} --warmup ${BENCH_WARMUP:-3} > $OUT/ab_${S}_flhs.json 2>> $OUT/ab.err; echo "LHS in the loop kernels, $S:"; summ $OUT/ab_${S}_flhs.json
  done; fi
if has abl2; then echo "== A/B: second fused kernel with the LHS inside at 5 wavefronts per SIMD (96 registers, spills) vs 4 (default)"
  for S in ${AB_SIZES:-256 512}; do
    for O in 1 0; do
    timeout 900 python bench.py --size $S --no-cpu --no-alt --no-pcie --no-checksum --debug-option loop2_flhs_five_waves=$O --steps ${BENCH_STEPS:-10} --warmup ${BENCH_WARMUP:-3} > $OUT/abl2_${S}_$O.json 2>> $OUT/ab.err; echo "five_waves=$O, $S:"; summ $OUT/abl2_${S}_$O.json
    done
  done; fi
if has tune; then echo "== tuning scan: wave priority while streaming (loop_prio = priority + 1), workgroups of the dot-product totals (sums_groups)"
  for S in ${AB_SIZES:-256 512}; do
    for O in "loop_prio=0" "loop_prio=2" "loop_prio=4" "sums_groups=256" "loop_prio=4 --debug-option sums_groups=256"; do
      timeout 900 python bench.py --size $S --no-cpu --no-alt --no-pcie --no-checksum --debug-option $O --steps ${BENCH_STEPS:-6} --warmup ${BENCH_WARMUP:-2} > $OUT/tune_${S}.json 2>> $OUT/ab.err; echo "$S $O:"; summ $OUT/tune_${S}.json | head -4 | grep -E "value|loop|finish"
      python - $OUT/tune_${S}.json "$S $O" >> $OUT/tune.jsonl <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {x["kernel"]: x["avg_ms"] for x in r["kernels"]}
print(json.dumps({"case": sys.argv[2], "ms_per_iteration": round(r["ms_per_step"] / r["config"]["bicgstab_iters_per_step"], 4), "loop1_cg": k.get("bicgstab_loop1_cg"), "loop2_cg": k.get("bicgstab_loop2_cg"), "dots_finish": k.get("bicgstab_dots_finish")}))
PY
    done
  done; cat $OUT/tune.jsonl; fi
if has nofuse; then echo "== A/B: host-driven unfused loops, 256^3"
  timeout 600 python bench.py --size 256 --no-cpu --no-alt --no-pcie --no-fuse --steps 5 --warmup 2 > $OUT/bench_256_nofuse.json 2>> $OUT/flavour.err; summ $OUT/bench_256_nofuse.json; fi
if has window256; then echo "== the driver's window at 256^3 (steps 26-45 after 5 warm-up steps) for profiles/r03/reference_window_256.json"
  timeout 900 python bench.py --size 256 --no-cpu --no-alt --no-pcie --steps 20 --warmup 5 > $OUT/bench_256_window.json 2> $OUT/bench_256_window.err ; echo "rc=$?"; summ $OUT/bench_256_window.json; fi
if has extras; then echo "== the other bench modes: stencil-only 512^3, multi-level mesh, implicit diffusion"
  timeout 600 python bench.py --stencil-only --no-cpu --steps 10 --warmup 3 > $OUT/bench_512_stencil_only.json 2> $OUT/extras.err; echo "stencil-only rc=$?"; summ $OUT/bench_512_stencil_only.json | head -5
  timeout 900 python bench.py --amr --steps 5 --warmup 2 > $OUT/bench_amr_3level.json 2>> $OUT/extras.err; echo "amr rc=$?"; summ $OUT/bench_amr_3level.json | head -6
  timeout 900 python bench.py --amr --block-solver 5 --steps 5 --warmup 2 > $OUT/bench_amr_3level_multigrid.json 2>> $OUT/extras.err; echo "amr multigrid rc=$?"; summ $OUT/bench_amr_3level_multigrid.json | head -6
  timeout 900 python bench.py --implicit-diffusion --no-cpu --steps 3 --warmup 1 > $OUT/bench_512_implicit.json 2>> $OUT/extras.err; echo "implicit rc=$?"; summ $OUT/bench_512_implicit.json | head -4
  tail -3 $OUT/extras.err; fi
if has full512; then echo "== the driver's command"
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_512_driver.json 2> $OUT/bench_512_driver.err ; echo "rc=$?"; summ $OUT/bench_512_driver.json; tail -3 $OUT/bench_512_driver.err; fi
ROOT=$PWD
if has trace; then echo "== rocprofv3 kernel trace of the bench command (512^3 full step, 5 steps)"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -o t -- python $ROOT/bench.py --no-cpu --no-alt --no-pcie --no-checksum --steps 5 --warmup 2 > $ROOT/$OUT/trace.log 2>&1 )
  for f in $(find $OUT/trace -name "*kernel_stats.csv" | head -1); do cp $f $OUT/rocprofv3_kernel_stats_512cubed_fullstep_bench.csv; head -14 $f | cut -c1-220; done
  grep -E "^\{" $OUT/trace.log | tail -c 600
  rm -rf $OUT/trace; fi
if has pmc; then echo "== rocprofv3 PMC passes (one pressure projection at 512^3): FETCH_SIZE, WRITE_SIZE in separate runs"
  for C in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$C -o p -- python $ROOT/scripts/kernel_probe.py one --size 512 --kernel solve > $ROOT/$OUT/pmc_$C.log 2>&1 )
    for f in $(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1); do python - "$f" $C <<'PY' | tee -a $OUT/pmc_solver_kernels_512cubed.txt
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
    rm -rf $OUT/pmc_$C
  done; fi
if has pmcsq; then echo "== rocprofv3 SQ counters of the fused kernels with the LHS inside (256^3, one pressure projection)"
  rm -f $OUT/pmc_fused_kernels_sq_256cubed.txt
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    TAGC=$(echo $SET | cut -d" " -f1)
    ( cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $ROOT/$OUT/pmc4_$TAGC -o p -- python $ROOT/scripts/kernel_probe.py one --size 256 --kernel solve > $ROOT/$OUT/pmc4_$TAGC.log 2>&1 )
    for f in $(find $OUT/pmc4_$TAGC -name "*counter_collection.csv" | head -1); do python - "$f" <<'PY' | tee -a $OUT/pmc_fused_kernels_sq_256cubed.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:52]
    if "k_loop1_cg" in k or "k_loop2_cg" in k or "k_precond<" in k or "k_lhs" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    for c, v in sorted(d.items()):
        print(k, c, "launches", len(v), "mean", round(sum(v) / len(v), 1))
PY
    done
    tail -2 $OUT/pmc4_$TAGC.log
    rm -rf $OUT/pmc4_$TAGC
  done; fi
echo "== done"
