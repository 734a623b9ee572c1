#!/usr/bin/env python3
"""Register / scratch table of every kernel in xgpr_amd/csrc/xgpr_hip.hip (gfx950, the flags of xgpr_amd/build.py).

    python tools/resource_usage.py [out.txt]          (default profiles/r5_resource_usage.txt)
    python tools/resource_usage.py --write-floor      (also rewrites tests/golden/resource_floor.json)

Parses hipcc's -Rpass-analysis=kernel-resource-usage remarks; no GPU needed.  tests/test_resource_usage.py runs
collect() and fails on any scratch and on any kernel whose register-limited occupancy fell below the committed floor."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOOR = os.path.join(ROOT, "tests", "golden", "resource_floor.json")
HEADER = ("# hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage xgpr_amd/csrc/xgpr_hip.hip (tools/resource_usage.py)\n"
          "# kernel | VGPRs | AGPRs | SGPRs | scratch bytes/lane | occupancy waves/SIMD (register-limited) | VGPR spill | SGPR spill\n")


def collect(extra_flags=()):
    """One dict per kernel: name, VGPRs, AGPRs, TotalSGPRs, ScratchSize, Occupancy, 'VGPRs Spill', 'SGPRs Spill'."""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-c", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", *extra_flags, os.path.join(ROOT, "xgpr_amd/csrc/xgpr_hip.hip"), "-o", os.devnull]
    err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    demangle = subprocess.run(["c++filt"], input=err, capture_output=True, text=True).stdout
    rows, cur = [], None
    for line in demangle.split("\n"):
        m = re.search(r"remark: .*Function Name: (.*?) \[-Rpass", line)
        if m:
            name = m.group(1)
            name = re.sub(r"^(void )?\(anonymous namespace\)::", "", name)
            name = re.sub(r"\([^()]*(\(anonymous namespace\)[^()]*)*\)$", "", name)
            cur = {"name": name}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


def write_table(rows, out_file):
    with open(out_file, "w") as f:
        f.write(HEADER)
        for r in rows:
            f.write(" | ".join(str(x) for x in (r["name"], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"), r.get("ScratchSize", "?"),
                                                r.get("Occupancy", "?"), r.get("VGPRs Spill", "?"), r.get("SGPRs Spill", "?"))) + "\n")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_file = args[0] if args else os.path.join(ROOT, "profiles", "r6_resource_usage.txt")
    rows = collect()
    write_table(rows, out_file)
    bad = [r for r in rows if r.get("ScratchSize", 0)]
    print(f"{len(rows)} kernels, {len(bad)} with scratch")
    for r in bad:
        print("  ", r["name"], "scratch", r["ScratchSize"], "VGPRs", r.get("VGPRs"), "spilled", r.get("VGPRs Spill"))
    if "--write-floor" in sys.argv:
        with open(FLOOR, "w") as f:
            json.dump({r["name"]: r["Occupancy"] for r in rows}, f, indent=0, sort_keys=True)
        print("wrote", FLOOR)


if __name__ == "__main__":
    main()
