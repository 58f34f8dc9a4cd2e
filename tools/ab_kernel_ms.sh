#!/bin/bash
# A/B of library builds on one box: tools/ab_kernel_ms.sh KERNEL_KEY lib_dir... — prints the step time and the named kernel's ms per step
# (bench.py's own per-kernel accounting) for the library in each directory (SCANRS_AMD_LIB).
KEY=$1; shift
for L in "$@"; do
  SCANRS_AMD_LIB=$PWD/$L/libscanrs_amd.so python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-heavy-tailed --no-randsvd --no-irlba --no-split-probe > /tmp/ab.json 2>/tmp/ab.err
  python3 - "$L" "$KEY" <<'PY'
import json, sys
d = json.loads([l for l in open("/tmp/ab.json") if l.startswith("{")][-1])
k = d["roofline"].get("kernel_ms_per_step", {})
print(sys.argv[1], "ms_per_step", d["ms_per_step"], {n: v for n, v in k.items() if sys.argv[2] in n})
PY
done
