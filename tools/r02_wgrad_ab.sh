#!/bin/bash
# A/B of the weight-gradient kernels on one box: conv parity tests, probe timings (streamed vs CRW_WGRAD=1 = first generation), full bench lines
set -o pipefail
O=$PWD/gpurun_out/r02w; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -q -m gpu -k "encoder or wgrad or full_model or trajectory" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?" | tee $O/status.txt
tail -4 $O/pytest_conv.log
for v in 0 1; do
  for c in "128 128" "64 128"; do
    CRW_WGRAD=$v timeout -k 10 120 python tools/probe_conv.py $c 3 16128 10 2>&1 | grep wgrad | sed "s/^/CRW_WGRAD=$v /" | tee -a $O/probe.log
  done
done
for v in 0 1 0 1; do
  CRW_WGRAD=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | grep '^{' > $O/bench_v$v.json
  python - <<PY
import json
d=json.load(open("$O/bench_v$v.json"))
print("CRW_WGRAD=$v ms_per_step", round(d["ms_per_step"],3), {k["kernel"]: round(k.get("launch_us",0)) for k in d.get("roofline_kernels",[]) if "wgrad" in k.get("kernel","")})
PY
done 2>&1 | tee -a $O/status.txt
