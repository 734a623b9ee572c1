#!/usr/bin/env python3
"""Instruction table of the fused CG matvec's per-datapoint loop, priced with measured issue costs.
    python tools/count_loop_insts.py [ztz3|conv] [LOG2P] [valu_cost.json] [out.json]
Compiles xgpr_amd/csrc/xgpr_hip.hip to gfx950 assembly (same flags as xgpr_amd/build.py), takes ztz3_kernel<LOG2P,
Z3_MATVEC> (per datapoint tile) or wave_conv_kernel<LOG2P, CONV_FGEN> (per k-mer tile), and counts the instructions
between the `XGPR_MARK loop_top` / `loop_end` comments, leaving out the regions between `cold_begin` / `cold_end` (the
large-argument cos/sin fix-up and the branch of a slot without a datapoint).  Each vector
instruction is priced with its class's issue cost from tools/valu_cost.hip (ns per wave-instruction per SIMD at THREE
waves per SIMD, the kernel's occupancy).  Result: profiles/r3_ztz3_inst_table.json, read by bench.py."""
import collections, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "ztz3"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else (10 if which == "ztz3" else 8)
cost_file = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r3_valu_cost.json")
out_file = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", f"r3_{which}_inst_table.json")
with tempfile.TemporaryDirectory() as td:
    asm = os.path.join(td, "x.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "xgpr_amd/csrc/xgpr_hip.hip"), "-o", asm], check=True, capture_output=True)
    lines = open(asm).read().split("\n")
name = f"ztz3_kernelILi{lg}ELi0ELb0EE" if which == "ztz3" else f"wave_conv_kernelILi{lg}ELi0E"      # MODE 0 = Z3_MATVEC / CONV_FGEN
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and name in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
body = lines[start:end]
marks = [(i, m.group(1)) for i, l in enumerate(body) for m in [re.search(r"XGPR_MARK (\w+)", l)] if m]
top = [i for i, m in marks if m == "loop_top"]; bot = [i for i, m in marks if m == "loop_end"]
assert len(top) == 1 and len(bot) == 1 and top[0] < bot[0], marks
cold, stack = [], []
for i, m in marks:
    if m == "cold_begin": stack.append(i)
    elif m == "cold_end": cold.append((stack.pop(), i))
counts = collections.Counter()
for i in range(top[0], bot[0]):
    if any(a <= i <= b for a, b in cold): continue
    l = body[i].strip()
    if not l or l[0] in ";." or l.endswith(":"): continue
    op = l.split()[0]
    if op == "v_pk_add_f32" and "op_sel" in l: op = "v_pk_add_f32 op_sel"
    counts[op] += 1
costs = json.load(open(cost_file))
alias = {"v_sub_f32": "v_add_f32", "v_fmac_f32": "v_fma_f32", "v_fma_f64": "v_fmac_f64", "v_mul_f64": "v_fmac_f64", "v_lshlrev_b32": "v_bitop3_b32",
         "v_and_b32": "v_add_u32", "v_xor_b32": "v_add_u32", "v_or_b32": "v_add_u32", "v_cndmask_b32": "v_cndmask_b32_e64", "v_mov_b64": "v_add_f64",
         "v_permlane16_swap_b32": "v_permlane32_swap_b32", "v_lshl_add_u32": "v_bitop3_b32", "v_add3_u32": "v_bitop3_b32", "v_readlane_b32": "v_mov_b32",
         "v_readfirstlane_b32": "v_mov_b32", "v_cmp_gt_i32": "v_cndmask_b32_e64", "v_cmp_ne_u32": "v_cndmask_b32_e64", "v_cmp_eq_u32": "v_cndmask_b32_e64", "v_cmp_ngt_f32": "v_cndmask_b32_e64", "v_cmp_lt_f32": "v_cndmask_b32_e64",
         "v_lshl_add_u64": "v_add_f64", "v_ashrrev_i32": "v_bitop3_b32"}
table, priced, unpriced = [], 0.0, []
for op, c in sorted(counts.items(), key=lambda kv: -kv[1]):
    base = re.sub(r"_e(32|64)$", "", op)
    key = base if base in costs else alias.get(base)
    kind = "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_")) else "salu"
    ns = costs[key]["W3"] if (kind == "valu" and key in costs) else None
    if kind == "valu" and ns is None: unpriced.append(op)
    if ns: priced += ns * c
    table.append({"op": op, "count": c, "class": kind, "ns_each_W3": ns, "ns_total": None if ns is None else round(ns * c, 1)})
res = {"kernel": f"ztz3_kernel<{lg}, Z3_MATVEC>" if which == "ztz3" else f"wave_conv_kernel<{lg}, CONV_FGEN>",
       "what": "instructions executed per wave per " + ("datapoint" if which == "ztz3" else "k-mer") + " tile (1024 frequencies) on the hot path of the main loop",
       "valu_instructions": sum(c for o, c in counts.items() if o.startswith("v_")),
       "lds_instructions": sum(c for o, c in counts.items() if o.startswith("ds_")),
       "other_instructions": sum(c for o, c in counts.items() if not o.startswith(("v_", "ds_"))),
       "priced_vector_ns_per_tile_per_simd": round(priced, 1), "unpriced_vector_ops": unpriced,
       "prices": os.path.relpath(cost_file, ROOT), "table": table}
json.dump(res, open(out_file, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "table"}, indent=1))
for r in table[:40]: print("%5d  %-28s %-5s %s" % (r["count"], r["op"], r["class"], r["ns_total"]))
