#!/usr/bin/env python3
"""Where a kernel's scratch instructions sit relative to the XGPR_MARK comments (loop_top / loop_end / cold_begin / cold_end).
    python tools/scratch_regions.py <mangled-name-fragment> ...     e.g. wave_conv_kernelILi8ELi0E"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "x.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "xgpr_amd/csrc/xgpr_hip.hip"), "-o", asm], check=True, capture_output=True)
    lines = open(asm).read().split("\n")
for name in sys.argv[1:]:
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and name in l and ":" in l.split(";")[0])
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
    region, c = "before loop_top", collections.Counter()
    for l in lines[start:end]:
        m = re.search(r"XGPR_MARK (\w+)", l)
        if m:
            region = "after " + m.group(1)
            continue
        s = l.strip()
        if s.startswith("scratch_"):
            c[(region, s.split()[0])] += 1
    print(name, dict(c) if c else "no scratch instructions")
