#!/bin/bash
# One GPU-box visit: smoke, parity tests, a short bench and a rocprofv3 kernel trace of it.
# Everything lands in gpurun_out/ (merged back by gpurun).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > $OUT/device.txt
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 $OUT/smoke.log
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -q -s -x --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -25 $OUT/pytest_gpu.log
if [ "$1" != "--tests-only" ]; then
echo "== bench" ; timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.log 2>&1 ; echo "bench rc=$?" ; tail -3 $OUT/bench.log
echo "== rocprofv3" ; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o cffm -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OLDPWD/$OUT/rocprof.log 2>&1) ; echo "rocprof rc=$?"
mkdir -p $OUT/prof ; find /tmp/prof -name "*stats*.csv" -exec cp {} $OUT/prof/ \; ; ls $OUT/prof | head
fi
