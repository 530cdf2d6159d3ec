#!/bin/bash
# One gpurun call of round 4.  Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round4.sh <tag> [stages...]'
TAG=${1:-r04a}; shift
STAGES=${@:-smoke constants newtests}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
has() { [[ " $STAGES " == *" $1 "* ]]; }
echo "== host: $(nproc) cpus, $(free -g | awk '/Mem:/{print $2}') GB, devices: $(python -c 'import cup3d_amd.capi as c; print(c.device_count())' 2>/dev/null)"
if has smoke; then echo "== smoke (release build)"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 $OUT/smoke.log; fi
if has constants; then echo "== device-side checksum constants (release build, one GPU)"
  timeout 600 python scripts/make_poisson_checksums.py ${CONST_SIZES:-64 128 256 512} --out $OUT/poisson_checksums.json --merge > $OUT/constants.log 2>&1 ; echo "constants rc=$?" ; tail -6 $OUT/constants.log; fi
if has newtests; then echo "== pytest: round-4 tests first"
  timeout 1200 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_release_flavour.py "tests/test_gpu_multirank.py::test_a_bad_call_on_one_rank_fails_on_every_rank_at_once" ${NEW_TESTS} -m gpu -q --durations=8 -s > $OUT/pytest_new.log 2>&1 ; echo "pytest rc=$?"
  grep -E "passed|failed|FAILED|Error|assert|ranks over" $OUT/pytest_new.log | tail -60 | cut -c1-600; fi
if has tests; then echo "== pytest -m gpu (everything)"
  timeout 1800 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; grep -E "passed|failed|FAILED|Error" $OUT/pytest_gpu.log | tail -40 | cut -c1-300; tail -12 $OUT/pytest_gpu.log | cut -c1-300; fi
if has resttests; then echo "== pytest -m gpu (all but the files of newtests)"
  timeout 1800 python -m pytest tests -m gpu -q --durations=10 --ignore=tests/test_gpu_rccl.py --ignore=tests/test_gpu_release_flavour.py > $OUT/pytest_rest.log 2>&1 ; echo "pytest rc=$?" ; grep -E "passed|failed|FAILED|Error" $OUT/pytest_rest.log | tail -40 | cut -c1-300; tail -12 $OUT/pytest_rest.log | cut -c1-300; fi
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
print("  comm", c.get("communication"))
print("  comm stream", r.get("communication_stream"))
for k in ("alt", "alt_multigrid", "alt_reference_association"):
    if r.get(k): print("  ", k, r[k].get("value"), r[k].get("bicgstab_iters_per_step"), r[k].get("ms_per_bicgstab_iteration"))
for k in r.get("kernels", [])[:7]:
    print("   ", k["kernel"], k["launches"], k["avg_ms"], k.get("frac"))
PY
}
for S in 128 256 512; do
  if has bench$S; then echo "== bench $S (release build, no cpu baseline, no alt)"
    timeout 900 python bench.py --size $S --no-cpu --no-alt --no-pcie --steps ${BENCH_STEPS:-10} --warmup ${BENCH_WARMUP:-3} ${BENCH_ARGS} > $OUT/bench_$S.json 2> $OUT/bench_$S.err ; echo "bench rc=$?" ; summ $OUT/bench_$S.json ; tail -2 $OUT/bench_$S.err
  fi
  for N in 2 4 8; do
    if has host$S.$N; then echo "== bench $S on $N ranks over the host-memory TEST transport"
      timeout 1200 python bench.py --gpus $N --transport host --size $S --no-cpu --no-alt --no-pcie --steps ${BENCH_STEPS:-5} --warmup ${BENCH_WARMUP:-2} > $OUT/bench_${S}_host_${N}ranks.json 2> $OUT/bench_${S}_host_${N}ranks.err ; echo "bench rc=$?" ; summ $OUT/bench_${S}_host_${N}ranks.json ; tail -2 $OUT/bench_${S}_host_${N}ranks.err
    fi
  done
done
if has driver; then echo "== the driver's command: python bench.py --gpus 1 --steps 20 --warmup 5"
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_512_fullstep.json 2> $OUT/bench_512_fullstep.err ; echo "bench rc=$?" ; summ $OUT/bench_512_fullstep.json ; tail -3 $OUT/bench_512_fullstep.err
fi
if has amr; then echo "== bench --amr (3 levels)"
  timeout 900 python bench.py --amr --steps ${AMR_STEPS:-10} --warmup 3 ${AMR_ARGS} > $OUT/bench_amr.json 2> $OUT/bench_amr.err ; echo "rc=$?" ; python - $OUT/bench_amr.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  value", r["value"], "ms/step", r["ms_per_step"], r["config"]["blocks"], r["config"]["bicgstab_iters_per_step"])
for k in r["kernels"][:10]: print("   ", k["kernel"], k["launches"], k["avg_ms"], k["share"])
PY
fi
if has configs4; then echo "== configs[4] at its stated size: reference CPU operators vs the drop-in (one rank)"
  timeout 1500 python scripts/configs4_measure.py --level-max ${C4_LEVEL:-7} --steps ${C4_STEPS:-5} --threads ${C4_THREADS:-32} --out $OUT/configs4_1024_effective.json > $OUT/configs4.log 2>&1 ; echo "configs4 rc=$?"
  python - $OUT/configs4_1024_effective.json <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
except Exception as e:
    print("  (no record)", e); sys.exit(0)
for k in ("blocks", "blocks_per_level", "hot_path_fraction_cpu", "hot_path_fraction_hip", "end_to_end_speedup", "hot_path_speedup", "pcie_MB_per_step", "block_lists_identical", "max_abs_dvel", "max_abs_vel"):
    print("  ", k, r.get(k))
print("   cpu", r["cpu"]["seconds_per_step"], r["cpu"]["per_operator_seconds_per_step"])
if "hip" in r: print("   hip", r["hip"]["seconds_per_step"], r["hip"]["per_operator_seconds_per_step"])
PY
  tail -3 $OUT/configs4.log | cut -c1-300
fi
if has amrab; then echo "== bench --amr A/B: every LHS a launch of its own (no_fuse_lhs_ml=1)"
  timeout 900 python bench.py --amr --steps ${AMR_STEPS:-10} --warmup 3 --debug-option no_fuse_lhs_ml=1 > $OUT/bench_amr_no_fuse_lhs_ml.json 2> $OUT/bench_amr_ab.err ; echo "rc=$?" ; python - $OUT/bench_amr_no_fuse_lhs_ml.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
its = r["config"]["bicgstab_iters_per_step"]
print("  value", r["value"], "ms/step", r["ms_per_step"], r["config"]["blocks"], its, "ms/iteration", round(r["ms_per_step"] / its, 4))
for k in r["kernels"][:8]: print("   ", k["kernel"], k["launches"], k["avg_ms"], k["share"])
PY
  echo "== the same with the testing build and the option off"
  timeout 900 python bench.py --amr --steps ${AMR_STEPS:-10} --warmup 3 --debug-option no_fuse_lhs_ml=0 > $OUT/bench_amr_fuse_lhs_ml.json 2>> $OUT/bench_amr_ab.err ; echo "rc=$?" ; python - $OUT/bench_amr_fuse_lhs_ml.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
its = r["config"]["bicgstab_iters_per_step"]
print("  value", r["value"], "ms/step", r["ms_per_step"], r["config"]["blocks"], its, "ms/iteration", round(r["ms_per_step"] / its, 4))
for k in r["kernels"][:8]: print("   ", k["kernel"], k["launches"], k["avg_ms"], k["share"])
PY
fi
if has amrbig; then echo "== bench --amr, four levels, >= 300 k blocks"
  timeout 1200 python bench.py --amr --amr-base ${AMRBIG_BASE:-4} --amr-levels 4 --amr-fraction ${AMRBIG_FRACTION:-0.45} --steps 5 --warmup 2 > $OUT/bench_amr_4level.json 2> $OUT/bench_amr_4level.err ; echo "rc=$?" ; python - $OUT/bench_amr_4level.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("  (no JSON)", e); sys.exit(0)
its = r["config"]["bicgstab_iters_per_step"]
print("  value", r["value"], "ms/step", r["ms_per_step"], r["config"]["blocks"], r["config"]["blocks_per_level"], its, "ms/iteration", round(r["ms_per_step"] / its, 4), "mesh build s", r["config"]["mesh_build_seconds"])
for k in r["kernels"][:10]: print("   ", k["kernel"], k["launches"], k["avg_ms"], k["share"])
PY
  tail -3 $OUT/bench_amr_4level.err
fi
if has micro; then echo "== SURVEY 8(d) micro-benchmarks at 512^3 (release build)"
  timeout 600 python bench.py --micro --steps 5 --no-cpu > $OUT/bench_512_micro.json 2> $OUT/bench_512_micro.err ; echo "rc=$?"
  python - $OUT/bench_512_micro.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for k in r["kernels"]: print("   ", k["kernel"], k["launches"], k["avg_ms"], k["frac"], "|", k["what"])
except Exception as e:
    print("  (no JSON)", e)
PY
  tail -2 $OUT/bench_512_micro.err; fi
if has stress; then echo "== SURVEY 8(d) solver-stress input (seeded random velocity) at 512^3"
  timeout 900 python bench.py --input random --steps 5 --warmup 2 --no-cpu --no-pcie > $OUT/bench_512_random_input.json 2> $OUT/bench_512_random_input.err ; echo "rc=$?" ; summ $OUT/bench_512_random_input.json ; tail -2 $OUT/bench_512_random_input.err; fi
if has pmc; then echo "== rocprofv3 PMC passes over the micro-benchmarks at 512^3: FETCH_SIZE, WRITE_SIZE in separate runs"
  ROOT=$(pwd); rm -f $OUT/pmc_poisson_kernels_512cubed.txt
  for CN in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 900 rocprofv3 --pmc $CN --kernel-trace --output-format csv -d $ROOT/$OUT/pmc_$CN -o p -- python $ROOT/bench.py --micro --steps 3 --no-cpu > $ROOT/$OUT/pmc_$CN.log 2>&1 )
    for f in $(find $OUT/pmc_$CN -name "*counter_collection.csv" | head -1); do python - "$f" $CN <<'PY' | tee -a $OUT/pmc_poisson_kernels_512cubed.txt
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") == c:
        acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    if any(t in k for t in ("k_loop", "k_lhs", "k_precond")): print(c, k, "launches", len(v), "mean_KB", round(sum(v) / len(v), 1))
PY
    done
    rm -rf $OUT/pmc_$CN
  done; fi
if has configs4mpi; then echo "== configs[4] at its stated size on 8 real MPI ranks through the shim (host-memory transport, one GPU)"
  CUP3D_CONFIGS4_LEVELMAX=7 timeout 2400 python -m pytest "tests/test_gpu_00_dropin_mpi.py" -m gpu -q -s -k configs4 > $OUT/pytest_configs4_levelmax7_8ranks.log 2>&1 ; echo "pytest rc=$?"
  grep -E "configs4|passed|failed|skipped|HUNG|Timeout" $OUT/pytest_configs4_levelmax7_8ranks.log | tail -8 | cut -c1-600; fi
if has ref256; then echo "== device vs the COMPILED REFERENCE at 256^3 periodic, three steps, default and tight tolerances"
  timeout 1500 python scripts/campaigns/baseline_sizes_vs_reference.py --size 256 --bc periodic --steps 3 --tight --threads ${REF_THREADS:-32} --out $OUT/reference_steps_256_periodic.json > $OUT/campaign256.log 2>&1 ; echo "rc=$?"; tail -c 1500 $OUT/campaign256.log; fi
if has ref512; then echo "== device vs the COMPILED REFERENCE at 512^3 all-wall, one step (minutes of host time)"
  timeout 2400 python scripts/campaigns/baseline_sizes_vs_reference.py --size 512 --bc wall --threads ${REF_THREADS:-32} --out $OUT/reference_step_512.json > $OUT/campaign512.log 2>&1 ; echo "rc=$?"; tail -c 1500 $OUT/campaign512.log; fi
if has mgab; then echo "== multigrid option at 512^3: smoother by one wavefront per block (production) vs workgroup per block (A/B), testing build"
  for V in 0 1; do
    timeout 600 python bench.py --block-solver 5 --steps 5 --warmup 2 --no-cpu --no-pcie --no-alt --debug-option mg_smooth_workgroup=$V > $OUT/bench_512_multigrid_smoother_$V.json 2> $OUT/bench_512_multigrid_smoother_$V.err ; echo "rc=$? (mg_smooth_workgroup=$V)"
    python - $OUT/bench_512_multigrid_smoother_$V.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  value", r["value"], "ms/step", r["ms_per_step"], "its/step", r["config"]["bicgstab_iters_per_step"])
for k in r["kernels"][:6]: print("   ", k["kernel"], k["launches"], k["avg_ms"], k["share"])
PY
  done; fi
if has mgtests; then echo "== pytest: the multigrid tests (one rank, over ranks, multi-level)"
  timeout 1200 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_amr.py tests/test_gpu_parity.py -m gpu -q -s -k "multigrid" --durations=5 > $OUT/pytest_multigrid.log 2>&1 ; echo "pytest rc=$?"
  grep -E "passed|failed|FAILED|Error|assert|iterations" $OUT/pytest_multigrid.log | tail -40 | cut -c1-400; fi
if has mgmpi; then echo "== configs[3] / configs[4] over real MPI ranks with the MULTIGRID preconditioner in the shim (CUP3D_HIP_BLOCK_SOLVER=5)"
  CUP3D_HIP_BLOCK_SOLVER=5 timeout 1500 python -m pytest tests/test_gpu_00_dropin_mpi.py -m gpu -q -s > $OUT/pytest_dropin_mpi_multigrid.log 2>&1 ; echo "pytest rc=$?"
  grep -E "configs|passed|failed|skipped|HUNG|Timeout|Error" $OUT/pytest_dropin_mpi_multigrid.log | tail -8 | cut -c1-600; fi
if has viewtests; then echo "== pytest: everything on rank views (thread ranks with poisoned ghost cells; real MPI ranks)"
  timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_00_dropin_mpi.py -m gpu -q -s --durations=5 > $OUT/pytest_views.log 2>&1 ; echo "pytest rc=$?"
  grep -E "passed|failed|FAILED|Error|assert|sub-boxes|configs" $OUT/pytest_views.log | tail -30 | cut -c1-400; fi
if has amrranks; then echo "== bench --amr over ranks (host-memory TEST transport, one GPU): sub-boxes vs whole ghost blocks; block CG and multigrid"
  for TAG2 in "subbox:" "whole:--debug-option whole_ghost_blocks=1" "multigrid:--block-solver 5"; do
    NAME=${TAG2%%:*}; EXTRA=${TAG2#*:}
    timeout 900 python bench.py --amr --gpus ${AMR_RANKS:-4} --transport host --steps 3 --warmup 1 $EXTRA > $OUT/bench_amr_${AMR_RANKS:-4}ranks_$NAME.json 2> $OUT/bench_amr_ranks_$NAME.err ; echo "rc=$? ($NAME)"
    python - $OUT/bench_amr_${AMR_RANKS:-4}ranks_$NAME.json <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
    c = r["config"]
    print("  value", r.get("value"), "blocks", c.get("blocks"), "its", c.get("bicgstab_iters_by_step"), "umax", [round(u, 9) for u in c.get("umax_by_step", [])])
    print("  comm", c.get("communication"))
except Exception as e:
    print("  (no JSON)", e, open(sys.argv[1]).read()[-600:])
PY
    tail -2 $OUT/bench_amr_ranks_$NAME.err | cut -c1-300
  done
  echo "-- the same on one rank"
  timeout 600 python bench.py --amr --steps 3 --warmup 1 > $OUT/bench_amr_1rank.json 2>/dev/null; python - $OUT/bench_amr_1rank.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = r["config"]
print("  value", r["value"], "blocks", c["blocks"], "its", c.get("bicgstab_iters_by_step"), "umax", [round(u, 9) for u in c.get("umax_by_step", [])])
PY
fi
if has tracealt; then echo "== rocprofv3 kernel trace of the direct-solve iteration (block_solver 1, the `alt` of the bench)"
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/tracealt -o alt -- python $OLDPWD/bench.py --block-solver 1 --steps 5 --warmup 2 --no-cpu --no-alt --no-pcie > $OLDPWD/$OUT/tracealt_bench.json 2> $OLDPWD/$OUT/tracealt.err ; echo "trace rc=$?"; cd $OLDPWD
  find $OUT/tracealt -name "*kernel_stats.csv" | head -1 | while read f; do head -6 "$f" | cut -c1-200; done
  find $OUT/tracealt -name "*kernel_trace.csv" -delete; find $OUT/tracealt -name "*.db" -delete
  summ $OUT/tracealt_bench.json
fi
if has fakerccl; then echo "== pytest: the RCCL code path with the stand-in library (bench over 2 / 3 processes; configs[3] through the shim on 2 MPI ranks)"
  CUP3D_TEST_RCCL_STAND_IN=1 timeout 1500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_00_dropin_mpi.py -m gpu -q -s -k "stand_in" --durations=5 > $OUT/pytest_rccl_stand_in.log 2>&1 ; echo "pytest rc=$?"
  grep -E "passed|failed|FAILED|Error|assert|ranks through|configs|fake_rccl" $OUT/pytest_rccl_stand_in.log | tail -30 | cut -c1-500; fi
if has fakeorder; then echo "== pytest: configs[3] over 2 MPI ranks, host transport THEN the stand-in (the order of the whole suite)"
  CUP3D_TEST_RCCL_STAND_IN=1 CUP3D_STALL_LIMIT=70 CUP3D_FAKE_RCCL_WAIT=40 timeout ${FAKEORDER_LIMIT:-280} python -m pytest tests/test_gpu_00_dropin_mpi.py -m gpu -q -s -x ${FAKEORDER_K:+-k "$FAKEORDER_K"} > $OUT/pytest_stand_in_after_host.log 2>&1 ; echo "pytest rc=$?"
  ls /dev/shm | head; df -h /dev/shm | tail -1
  grep -c "REF alive" $OUT/pytest_stand_in_after_host.log
  grep -E "passed|failed|FAILED|Error|assert|fake_rccl|HUNG|configs" $OUT/pytest_stand_in_after_host.log | tail -30 | cut -c1-1200; fi
if has trace; then echo "== rocprofv3 kernel trace of the driver's bench"
  cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/trace -o fullstep -- python $OLDPWD/bench.py --steps ${TRACE_STEPS:-20} --warmup 5 --no-cpu --no-alt --no-pcie > $OLDPWD/$OUT/trace_bench.json 2> $OLDPWD/$OUT/trace.err ; echo "trace rc=$?"; cd $OLDPWD
  find $OUT/trace -name "*kernel_stats.csv" | head -2 | while read f; do head -12 "$f" | cut -c1-220; done
  find $OUT/trace -name "*kernel_trace.csv" -delete; find $OUT/trace -name "*.db" -delete
fi
ls -la $OUT | head -30
