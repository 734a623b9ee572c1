#!/usr/bin/env python3
"""Source lint: no inline-assembly statement under xgpr_amd/csrc/ may load into a REGISTER.

A load issued from inline assembly returns before its data has landed; the compiler does not know that, so it is free to
copy or spill the destination register ahead of the s_waitcnt the author wrote in a later statement (this happened in
wave_conv_kernel in rounds 2-3: a stale window once in ~700 launches under load).  Loads the compiler can see
(ordinary pointers, __builtin_amdgcn_raw_buffer_load_*) carry their own wait counts.  Allowed from assembly: the LDS-DMA
forms (global_load_lds_* / buffer_load_* ... lds), which write LDS and are waited for with a counted vmcnt in front of a
barrier.

    python tools/lint_asm_loads.py        exit code 1 and one line per offence
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xgpr_amd", "csrc")
LOAD = re.compile(r"\b((?:global|flat|scratch|buffer|tbuffer)_load_\w+|s_(?:buffer_)?load_\w+|ds_(?:read|load|bpermute|permute|swizzle)\w*)\b")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def asm_statements(text):
    """(line number, statement text) of every asm(...) / asm volatile(...) statement."""
    for m in re.finditer(r"\basm\s*(?:volatile\s*)?\(", text):
        depth, i = 1, m.end()
        while i < len(text) and depth:
            c = text[i]
            if c == '"':                                  # skip string literals
                i += 1
                while text[i] != '"':
                    i += 2 if text[i] == "\\" else 1
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            i += 1
        yield text.count("\n", 0, m.start()) + 1, text[m.start():i]


def offences(path):
    out = []
    for line, stmt in asm_statements(strip_comments(open(path).read())):
        code = " ".join(re.findall(r'"((?:[^"\\]|\\.)*)"', stmt))
        for inst in re.sub(r"\\[nt]", "\n", code).split("\n"):
            m = LOAD.search(inst)
            if not m:
                continue
            lds_dma = "_load_lds_" in m.group(1) or re.search(r"\blds\b", inst)
            if not lds_dma:
                out.append((path, line, m.group(1)))
    return out


def main():
    bad = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".inc", ".hip", ".h")):
            bad += offences(os.path.join(CSRC, f))
    for path, line, inst in bad:
        print(f"{os.path.relpath(path, ROOT)}:{line}: inline-assembly load into a register: {inst}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
