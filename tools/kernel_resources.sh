#!/bin/bash
# registers, spills, LDS of the kernels of one translation unit of the product library (from the code object's metadata):
#   tools/kernel_resources.sh [object = disco_engine] [name pattern]
obj=sasktran2_amd/_lib/${1:-disco_engine}.o
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $obj $tmp/fatbin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fatbin --output=$tmp/co.o --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/co.o > $tmp/notes.txt
python3 - "$tmp/notes.txt" "$2" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    rows.append((g("name"), blk.split()[0], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
    n = n.split("(")[0].replace("void skdev::", "").replace("skd::", "")
    if pat and pat not in n:
        continue
    print(f"{n[:64]:64s} vgpr {r[2]:>4s} agpr {r[1]:>4s} sgpr {r[3]:>4s} spilled {r[4]:>4s} lds {r[5]:>6s} scratch {r[6]:>5s}")
PY
rm -rf $tmp
