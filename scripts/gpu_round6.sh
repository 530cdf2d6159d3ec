#!/bin/bash
# One gpurun call of round 6.  Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round6.sh <tag> [stages...]'
# Every pytest stage is self-describing even when it is cut off: PYTHONFAULTHANDLER, -rA --tb=long, and tests/conftest.py's live log
# (CUP3D_LIVE_LOG: start / outcome of every test and a failure's traceback at once, fsync'ed).
TAG=${1:-r06a}; shift
STAGES=${@:-smoke suite}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONFAULTHANDLER=1 PYTHONUNBUFFERED=1
cd "$(dirname "$0")/.."
has() { [[ " $STAGES " == *" $1 "* ]]; }
echo "== host: $(nproc) cpus, $(free -g | awk '/Mem:/{print $2}') GB, devices: $(python -c 'import cup3d_amd.capi as c; print(c.device_count())' 2>/dev/null), $(date)"
pt() {  # pt <log name> <limit s> <pytest args...>
  local name=$1 limit=$2; shift 2
  CUP3D_LIVE_LOG=$OUT/$name.live.log timeout $limit stdbuf -oL -eL python -m pytest "$@" -m gpu -q -rA --tb=long -o log_cli=false --durations=12 > $OUT/$name.log 2>&1
  local rc=$?
  echo "pytest [$name] rc=$rc"; grep -E "^(FAILED|ERROR)|passed|failed| error" $OUT/$name.log | tail -25 | cut -c1-400
  if [ $rc -ne 0 ]; then echo "--- last lines of the live log:"; tail -15 $OUT/$name.live.log | cut -c1-600; fi
  return $rc
}
summ() { python - "$1" <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print("  (no JSON)", e); sys.exit(0)
if r.get("valid") is False and r.get("value") is None:
    print("  ERROR LINE:", json.dumps(r)[:1500]); sys.exit(0)
c = r["config"]
print("  value", r["value"], "ms/step", r["ms_per_step"], "its/step", c.get("bicgstab_iters_per_step"), "ms/iteration", r.get("ms_per_bicgstab_iteration"), "lib", c.get("library"))
ck = c.get("checksum") or {}
print("  checksum", ck.get("ok"), {k: ck[k].get("ok") for k in ck if isinstance(ck[k], dict)}, "unchecked", ck.get("unchecked"))
cm = c.get("communication") or {}
print("  comm", {k: cm.get(k) for k in ("rccl_ranks", "allreduce_ms_per_iteration", "exposed_scalar_wait_ms_per_iteration", "exposed_halo_wait_ms_per_iteration", "halo_ms_per_iteration", "host_wait_fraction")})
if r.get("stencil_only"): print("  stencil_only", {k: v for k, v in r["stencil_only"].items() if k not in ("what", "target")})
for k in ("alt", "alt_multigrid", "alt_reference_association"):
    if r.get(k): print("  ", k, r[k].get("value"), r[k].get("bicgstab_iters_per_step"), r[k].get("ms_per_bicgstab_iteration"))
for k in (r.get("alt_multigrid") or {}).get("kernels", [])[:12]:
    print("    mg", k["kernel"], "L%d" % k["level"], k["blocks"], k["launches"], k["avg_ms"], k["frac"])
for k in r.get("kernels", [])[:9]:
    print("   ", k["kernel"], k["launches"], k["avg_ms"], k.get("frac"), k.get("share"))
PY
}
if has smoke; then echo "== smoke (release build)"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 $OUT/smoke.log; fi
if has standin; then echo "== the RCCL stand-in cases alone (stream-memory-operation mode + the host-function variant)"
  pt pytest_stand_in 1500 tests/test_gpu_rccl.py -k "stand_in" ; pt pytest_stand_in_mpi 1500 tests/test_gpu_00_dropin_mpi.py -k "rccl_stand_in" -s; fi
if has quick; then echo "== pytest: ${QUICK}"
  pt pytest_quick ${QUICK_LIMIT:-1200} ${QUICK} ; fi
if has exitdiag; then echo "== does the test process leave cleanly?  (glibc checks every free: MALLOC_CHECK_=3)"
  for SEL in "fused_refresh" "not fused_refresh"; do
    MALLOC_CHECK_=3 timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_resources.py -m gpu -q -k "$SEL" > "$OUT/exitdiag_${SEL// /_}.log" 2>&1 ; echo "exit code [$SEL] = $?"
    tail -4 "$OUT/exitdiag_${SEL// /_}.log" | cut -c1-300
  done
  timeout 600 python -X faulthandler -m pytest tests/test_gpu_resources.py -m gpu -q > $OUT/exitdiag_resources_alone.log 2>&1 ; echo "exit code [resources alone, no malloc check] = $?"; tail -3 $OUT/exitdiag_resources_alone.log | cut -c1-300
fi
if has earlydet; then echo "== is the early all-reduce deterministic?  the same forced-communicator run ${DET_RUNS:-5} times at 128^3 (and at 64^3): iteration counts"
  for SZ in 128 64; do for R in $(seq 1 ${DET_RUNS:-5}); do
    CUP3D_RCCL_LIBRARY=$PWD/tests/fake_rccl/librccl_fake.so FAKE_RCCL_ALLREDUCE_US=30 CUP3D_FORCE_COMM=1 timeout 300 python bench.py --full-line --detail-out '' --size $SZ --steps 3 --warmup 1 --no-cpu --no-alt --no-pcie \
      --debug-option force_allreduce=1 --debug-option early_allreduce=1 ${DET_ARGS} 2> $OUT/earlydet_${SZ}_$R.err | python -c "import sys, json; r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $SZ run $R', r['config']['bicgstab_iters_by_step'], r['config']['umax_by_step'][-1])"
  done; done; fi
if has suite; then echo "== pytest -m gpu (the WHOLE suite, stand-in cases included)"
  pt pytest_gpu ${SUITE_LIMIT:-2400} tests ; tail -5 $OUT/pytest_gpu.log | cut -c1-300; fi
for S in 128 256 512; do
  if has bench$S; then echo "== bench $S (release build, no cpu baseline, no alt)"
    timeout 900 python bench.py --full-line --detail-out '' --size $S --no-cpu --no-alt --no-pcie --steps ${BENCH_STEPS:-10} --warmup ${BENCH_WARMUP:-3} ${BENCH_ARGS} > $OUT/bench_$S.json 2> $OUT/bench_$S.err ; echo "bench rc=$?" ; summ $OUT/bench_$S.json ; tail -2 $OUT/bench_$S.err
  fi
done
if has ab256; then echo "== A/B at 256^3 (one GPU's share of the headline on 8): default | no per-kernel events"
  for V in "default:" "no_profile:--no-profile"; do
    N=${V%%:*}; A=${V#*:}
    timeout 600 python bench.py --full-line --detail-out '' --size ${AB_SIZE:-256} --no-cpu --no-alt --no-pcie --steps ${AB_STEPS:-10} --warmup 3 $A > $OUT/bench_${AB_SIZE:-256}_$N.json 2> $OUT/bench_${AB_SIZE:-256}_$N.err ; echo "rc=$? ($N)"; summ $OUT/bench_${AB_SIZE:-256}_$N.json | head -1; summ $OUT/bench_${AB_SIZE:-256}_$N.json | grep bicgstab_loop
  done; fi
if has driver; then echo "== the driver's command: python bench.py --gpus 1 --steps 20 --warmup 5 (stdout = the compact line; full record = bench_detail.json)"
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_512_driver_stdout.json 2> $OUT/bench_512_fullstep.err ; echo "bench rc=$?"
  echo "  stdout: $(wc -l < $OUT/bench_512_driver_stdout.json) line(s), $(wc -c < $OUT/bench_512_driver_stdout.json) bytes"
  python -c "import json,sys; r=json.loads(open('$OUT/bench_512_driver_stdout.json').read().strip().splitlines()[-1]); print('  parsed: value', r['value'], 'ms/it', r.get('ms_per_bicgstab_iteration'), 'roofline', r['roofline'], 'cpu_baseline', r.get('cpu_baseline'), 'stencil_only', r.get('stencil_only'))"
  cp gpurun_out/bench_detail.json $OUT/bench_512_fullstep.json 2>/dev/null || cp bench_detail.json $OUT/bench_512_fullstep.json
  summ $OUT/bench_512_fullstep.json ; tail -3 $OUT/bench_512_fullstep.err
fi
if has exitrepro; then echo "== exit-time double free: minimal reproducer, order of {library RCCL communicator, torch}; exit codes under MALLOC_CHECK_=3"
  for ORD in ${EXIT_ORDERS:-rccl,torch torch,rccl rccl torch sim,torch rccl,sim,torch rcclkeep,torch torch,rcclkeep}; do
    MALLOC_CHECK_=3 timeout 300 python -X faulthandler scripts/exit_repro.py $ORD > $OUT/exitrepro_${ORD//,/_}.log 2>&1 ; echo "exit code [$ORD] = $?"
    grep -E "mapped|double free|corruption|Abort" $OUT/exitrepro_${ORD//,/_}.log | cut -c1-500
  done
  echo "-- LD_PRELOAD of the system librccl, order rccl,torch"
  LD_PRELOAD=/opt/rocm/lib/librccl.so.1 MALLOC_CHECK_=3 timeout 300 python -X faulthandler scripts/exit_repro.py rccl,torch > $OUT/exitrepro_preload.log 2>&1 ; echo "exit code [preload rccl,torch] = $?"
  grep -E "mapped|double free|corruption" $OUT/exitrepro_preload.log | cut -c1-500
fi
if has exitsubset; then echo "== exit-time double free: the two-file subset of round 5, as is / without the RCCL plumbing test / torch imported first"
  MALLOC_CHECK_=3 timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_resources.py -m gpu -q -x > $OUT/exitsubset_as_is.log 2>&1 ; echo "exit code [as is] = $?"; tail -3 $OUT/exitsubset_as_is.log | cut -c1-300
  if [ -z "$EXITSUBSET_ONLY_AS_IS" ]; then
  MALLOC_CHECK_=3 timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_resources.py -m gpu -q -x -k "not rccl_plumbing" > $OUT/exitsubset_no_plumbing.log 2>&1 ; echo "exit code [no plumbing test] = $?"; tail -3 $OUT/exitsubset_no_plumbing.log | cut -c1-300
  CUP3D_TEST_TORCH_FIRST=1 MALLOC_CHECK_=3 timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_resources.py -m gpu -q -x > $OUT/exitsubset_torch_first.log 2>&1 ; echo "exit code [torch first] = $?"; tail -3 $OUT/exitsubset_torch_first.log | cut -c1-300
  fi
fi
if has refexit; then echo "== C++ host: the unmodified reference TU + the shim leaves with exit code 0 under MALLOC_CHECK_=3"
  MALLOC_CHECK_=3 timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x -k "${REFEXIT_K:-exit}" > $OUT/refexit.log 2>&1; echo "rc=$?"; tail -3 $OUT/refexit.log | cut -c1-300
fi
if has driverq; then echo "== the driver's workload, shorter: --steps 8 --warmup 3, no CPU baseline"
  timeout 1200 python bench.py --full-line --detail-out '' --gpus 1 --steps 8 --warmup 3 --no-cpu ${DRIVERQ_ARGS} > $OUT/bench_512_short.json 2> $OUT/bench_512_short.err ; echo "bench rc=$?" ; summ $OUT/bench_512_short.json ; tail -3 $OUT/bench_512_short.err
fi
if has amr; then echo "== bench --amr (3 levels)"
  timeout 900 python bench.py --full-line --detail-out '' --amr --steps ${AMR_STEPS:-10} --warmup 3 ${AMR_ARGS} > $OUT/bench_amr.json 2> $OUT/bench_amr.err ; echo "rc=$?" ; python - $OUT/bench_amr.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
its = r["config"]["bicgstab_iters_per_step"]
print("  value", r["value"], "ms/step", r["ms_per_step"], r["config"]["blocks"], its, "ms/iteration", round(r["ms_per_step"] / its, 4))
for k in r["kernels"][:10]: print("   ", k["kernel"], k["launches"], k["avg_ms"], k["share"])
PY
fi
if has amrab; then echo "== bench --amr A/B (testing build, same box): the flux correction as ONE launch (production) vs one launch per direction (rounds 1-4)"
  for V in 0 1; do
    timeout 600 python bench.py --full-line --detail-out '' --amr --steps ${AMR_STEPS:-10} --warmup 3 --debug-option flux_fix_by_direction=$V > $OUT/bench_amr_flux_fix_by_direction_$V.json 2> $OUT/bench_amr_ab_$V.err ; echo "rc=$? (flux_fix_by_direction=$V)"; python - $OUT/bench_amr_flux_fix_by_direction_$V.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
its = r["config"]["bicgstab_iters_per_step"]
print("  value", r["value"], "ms/step", r["ms_per_step"], r["config"]["blocks"], its, "ms/iteration", round(r["ms_per_step"] / its, 4))
for k in r["kernels"][:9]: print("   ", k["kernel"], k["launches"], k["avg_ms"], k["share"])
PY
  done; fi
# injected latency: what an iteration costs when every all-reduce takes L microseconds (FAKE_RCCL_ALLREDUCE_US), ONE process whose scalars
# are forced through the communicator (force_allreduce: testing build), so that nothing but the latency changes between the runs
if has latency; then echo "== injected all-reduce latency, one process, ${LAT_SIZE:-256}^3: ms per BiCGSTAB iteration and the exposed scalar wait"
  for MODE in ${LAT_MODES:-0 1}; do for L in ${LAT_US:-0 25 50 100}; do
    F=$OUT/bench_${LAT_SIZE:-256}_latency_${L}us_early${MODE}.json
    CUP3D_RCCL_LIBRARY=$PWD/tests/fake_rccl/librccl_fake.so FAKE_RCCL_ALLREDUCE_US=$L CUP3D_FORCE_COMM=1 timeout 600 python bench.py --full-line --detail-out '' --size ${LAT_SIZE:-256} --no-cpu --no-alt --no-pcie --steps ${LAT_STEPS:-6} --warmup 2 \
      --debug-option force_allreduce=1 --debug-option early_allreduce=$MODE > $F 2> ${F%.json}.err ; echo "rc=$? (latency $L us, early_allreduce=$MODE)"; summ $F | head -4; tail -2 ${F%.json}.err
  done; done; fi
if has latency2; then echo "== injected all-reduce latency, TWO processes on one GPU (bench.py --gpus 2, stand-in library), ${LAT_SIZE:-256}^3"
  for MODE in ${LAT_MODES:-0 1}; do for L in ${LAT_US:-0 50}; do
    F=$OUT/bench_${LAT_SIZE:-256}_2ranks_latency_${L}us_early${MODE}.json
    CUP3D_RCCL_LIBRARY=$PWD/tests/fake_rccl/librccl_fake.so CUP3D_BENCH_SHARE_DEVICE=1 FAKE_RCCL_ALLREDUCE_US=$L CUP3D_EARLY_ALLREDUCE=$MODE timeout 900 python bench.py --full-line --detail-out '' --gpus 2 --size ${LAT_SIZE:-256} --no-cpu --no-alt --no-pcie \
      --steps ${LAT_STEPS:-4} --warmup 2 > $F 2> ${F%.json}.err ; echo "rc=$? (2 ranks, latency $L us, early=$MODE)"; summ $F | head -4; tail -2 ${F%.json}.err
  done; done; fi
if has trace; then echo "== rocprofv3 --kernel-trace --stats of the driver's workload (short)"
  ROOT=$(pwd); ( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/trace -o t -- python $ROOT/bench.py --steps ${TRACE_STEPS:-6} --warmup 2 --no-cpu --no-pcie ${TRACE_ARGS} > $ROOT/$OUT/trace.log 2>&1 )
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/rocprofv3_kernel_stats.csv && head -25 $f | cut -c1-200; rm -rf $OUT/trace; fi
if has pmcmg; then echo "== rocprofv3 PMC passes over the multigrid option at 512^3 (FETCH_SIZE, WRITE_SIZE in separate runs)"
  ROOT=$(pwd); rm -f $OUT/pmc_multigrid_kernels_512cubed.txt
  for CN in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 900 rocprofv3 --pmc $CN --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$CN -o p -- python $ROOT/bench.py --block-solver 5 --steps 2 --warmup 1 --no-cpu --no-alt --no-pcie --no-checksum > $ROOT/$OUT/pmcmg_$CN.log 2>&1 )
    for f in $(find $OUT/pmc_$CN -name "*counter_collection.csv" | head -1); do python scripts/pmc_by_kernel_and_grid.py "$f" $CN k_mg_ k_advdiff k_solver_init k_copy k_lhs | tee -a $OUT/pmc_multigrid_kernels_512cubed.txt; done
    rm -rf $OUT/pmc_$CN
  done; fi
if has pmcmain; then echo "== rocprofv3 PMC passes over the driver's workload at 512^3, 2 steps (FETCH_SIZE, WRITE_SIZE in separate runs)"
  ROOT=$(pwd); rm -f $OUT/pmc_fullstep_kernels_512cubed.txt
  for CN in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 900 rocprofv3 --pmc $CN --kernel-trace --output-format csv -d $ROOT/$OUT/pmcm_$CN -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-alt --no-pcie --no-checksum > $ROOT/$OUT/pmcmain_$CN.log 2>&1 )
    for f in $(find $OUT/pmcm_$CN -name "*counter_collection.csv" | head -1); do python scripts/pmc_by_kernel_and_grid.py "$f" $CN k_loop k_advdiff k_lhs k_precond k_refresh k_solver_init k_copy | tee -a $OUT/pmc_fullstep_kernels_512cubed.txt; done
    rm -rf $OUT/pmcm_$CN
  done; fi
if has cgcg; then echo "== EXPERIMENT: single-reduction (Chronopoulos-Gear) block CG, EV 32: stand-alone kernel, then behind the loops"
  for SZ in 256 512; do CUP3D_HIP_FLAVOUR=testing timeout 600 python scripts/kernel_probe.py cgvar --size $SZ --variants 0,32 --reps 6 2>&1 | grep probe | tee -a $OUT/probe_block_cg_single_reduction.jsonl; done
  for SZ in ${CGCG_SIZES:-256 512}; do for V in 0 40; do
    F=$OUT/bench_${SZ}_cg_variant_$V.json
    timeout 900 python bench.py --full-line --detail-out '' --size $SZ --no-cpu --no-alt --no-pcie --steps ${CGCG_STEPS:-8} --warmup 3 --debug-option cg_variant=$V > $F 2> ${F%.json}.err; echo "rc=$? (size $SZ cg_variant $V)"; summ $F | head -1; summ $F | grep bicgstab_loop; tail -1 ${F%.json}.err
  done; done
fi
if has fdmband; then echo "== iteration counts of the fused / unfused direct block solve (ADVICE r5: is the 1.5x band still needed?)"
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "direct_block_solve_inside" 2>&1 | grep -E "fused / unfused|passed|failed" | tee $OUT/fdm_counts.log
fi
if has stencil256; then echo "== BASELINE configs[1] at its own size: 256^3 periodic Taylor-Green, advect-diffuse only (+ the reference operator on the host)"
  for SZ in ${STENCIL_SIZES:-256}; do
    timeout 900 python bench.py --stencil-only --size $SZ --steps ${STENCIL_STEPS:-100} --warmup 10 --detail-out bench_detail_stencil_$SZ.json > $OUT/bench_${SZ}_stencil_only_stdout.json 2> $OUT/bench_${SZ}_stencil_only.err; echo "rc=$?"
    cp gpurun_out/bench_detail_stencil_$SZ.json $OUT/bench_${SZ}_stencil_only.json; cat $OUT/bench_${SZ}_stencil_only_stdout.json | cut -c1-3000; tail -2 $OUT/bench_${SZ}_stencil_only.err
  done
fi
if has slope; then echo "== EXPERIMENT: what does the BiCGSTAB iteration pay per byte?  dummy streams ADDED to the production loop kernels (+16 / +32 B per cell and iteration)"
  for SZ in ${SLOPE_SIZES:-512 256}; do for V in 0 1 2; do
    F=$OUT/bench_${SZ}_extra_streams_$V.json
    timeout 900 python bench.py --full-line --detail-out '' --size $SZ --no-cpu --no-alt --no-pcie --steps ${SLOPE_STEPS:-8} --warmup 3 --debug-option extra_streams=$V > $F 2> ${F%.json}.err; echo "rc=$? (size $SZ extra_streams $V)"; summ $F | head -1; summ $F | grep bicgstab_loop; tail -1 ${F%.json}.err
  done; done
fi
if has haloprobe; then echo "== halo overlap probe (two thread ranks, injected slab latency), ${HALO_RUNS:-3} runs"
  for R in $(seq 1 ${HALO_RUNS:-3}); do timeout 600 python scripts/halo_overlap_probe.py --delays ${HALO_DELAYS:-0,50,100,200,400,800} --iters 40 2>> $OUT/halo_probe.err > $OUT/halo_overlap_probe_run$R.jsonl; python - $OUT/halo_overlap_probe_run$R.jsonl <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = None
for r in rows:
    e = [v for k, v in r.items() if k.startswith("exposed_halo_wait")][0]
    base = e if base is None else base
    inj = r["injected_ms_per_iteration"]
    print("  delay", r["injected_delay_us_per_exchange"], "us: wall", r["wall_ms_per_iteration"], "exposed", e, "injected", inj, "incremental exposed/injected", round((e - base) / inj, 3) if inj else None)
PY
  done; fi
if has cgev; then echo "== the evaluation variants of the block CG BEHIND THE LOOPS on the real workload (fused_cg_ev: 2 = single-width LDS reads, 8 = three-operand FMA p update, 4 = reciprocal divisions)"
  for SZ in ${CGEV_SIZES:-512 256}; do for V in ${CGEV_VARIANTS:-0 2 8 10 14 0}; do
    F=$OUT/bench_${SZ}_fused_cg_ev_$V.json
    timeout 900 python bench.py --full-line --detail-out '' --size $SZ --no-cpu --no-alt --no-pcie --steps ${CGEV_STEPS:-8} --warmup 3 --debug-option fused_cg_ev=$V > $F 2> ${F%.json}.err; echo "rc=$? (size $SZ fused_cg_ev $V)"; summ $F | head -1; summ $F | grep bicgstab_loop; tail -1 ${F%.json}.err | grep -v amdgpu.ids
  done; done
fi
if has advvar; then echo "== advect-diffuse stage with single-width LDS reads (advdiff_variant 7) against the production kernel (16 = 0 through the testing build)"
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "kernel_variants_bit_exact" 2>&1 | tail -2
  for SZ in ${ADV_SIZES:-512 256}; do for V in ${ADV_VARIANTS:-16 7 16 7}; do
    F=$OUT/bench_${SZ}_stencil_only_advdiff_variant_$V.json
    timeout 600 python bench.py --full-line --detail-out '' --stencil-only --size $SZ --no-cpu --steps ${ADV_STEPS:-60} --warmup 10 --debug-option advdiff_variant=$V > $F 2> ${F%.json}.err; echo "rc=$? (size $SZ advdiff_variant $V)"
    python -c "import json; r=json.loads(open('$F').read().strip().splitlines()[-1]); print('  value', r['value'], 'ms/step', r['ms_per_step'], [(k['kernel'], k['avg_ms'], k.get('frac')) for k in r['kernels'][:3]])"
  done; done
fi
if has opt; then echo "== A/B of one debug option on the real workload: OPT_NAME=${OPT_NAME} values ${OPT_VALUES:-0 1 0 1}"
  for SZ in ${OPT_SIZES:-512 256}; do for V in ${OPT_VALUES:-0 1 0 1}; do
    F=$OUT/bench_${SZ}_${OPT_NAME}_$V.json
    timeout 900 python bench.py --full-line --detail-out '' --size $SZ --no-cpu --no-alt --no-pcie --steps ${OPT_STEPS:-8} --warmup 3 --debug-option ${OPT_NAME}=$V > $F 2> ${F%.json}.err; echo "rc=$? (size $SZ ${OPT_NAME} $V)"; summ $F | head -1; summ $F | grep bicgstab_loop; tail -1 ${F%.json}.err | grep -v amdgpu.ids
  done; done
fi
echo "== done $(date)"
