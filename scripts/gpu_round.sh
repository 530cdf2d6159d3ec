#!/bin/bash
# One gpurun call: smoke, GPU parity tests, bench lines, rocprof kernel trace.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
ROOT=$PWD
echo "== rocminfo" ; rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" ; nproc ; free -g | head -2
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 $OUT/smoke.log
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -15 $OUT/pytest_gpu.log
echo "== bench 256 stencil-only"
timeout 300 python bench.py --size 256 --steps 10 --warmup 2 --no-cpu --stencil-only > $OUT/bench_256_stencil.json 2> $OUT/bench_256_stencil.err ; tail -c 1500 $OUT/bench_256_stencil.json
echo "== bench 512 stencil-only"
timeout 300 python bench.py --size 512 --steps 5 --warmup 1 --no-cpu --stencil-only > $OUT/bench_512_stencil.json 2> $OUT/bench_512_stencil.err ; tail -c 1500 $OUT/bench_512_stencil.json
echo "== bench default (512 full + cpu baseline)"
timeout 900 python bench.py > $OUT/bench_512.json 2> $OUT/bench_512.err ; echo "bench rc=$?" ; tail -c 3000 $OUT/bench_512.json ; tail -5 $OUT/bench_512.err
echo "== rocprof kernel trace (256 full, 2 steps)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o trace -- python $ROOT/bench.py --size 256 --steps 2 --warmup 1 --no-cpu > $ROOT/$OUT/rocprof.log 2>&1 )
find $OUT/prof -name "*stats*" | head ; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -25 $f; done
echo "== done"
