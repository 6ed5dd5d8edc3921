#!/bin/bash
# register / spill report of the kernels in one translation unit: regs.sh als_cg_q [filter]
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-pass-failed $EXTRA --cuda-device-only -S -o build/$1.s implicit_amd/csrc/$1.hip 2>&1 | grep -i "error" 
python - "$1" "${2:-}" <<'PY'
import re,sys
s=open(f"build/{sys.argv[1]}.s").read()
for m in re.finditer(r"\.name:\s+(\S+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", s, re.S):
    name=m.group(1)
    if sys.argv[2] in name:
        import subprocess
        d=subprocess.run(["c++filt", name],capture_output=True,text=True).stdout.strip()
        print(f"vgpr {m.group(3):>4} spill {m.group(4):>4} sgpr {m.group(2):>4}  {d[:110]}")
PY
