#!/usr/bin/env python3
"""Instruction-class sequence of a kernel's hottest loop from a -save-temps .s file:  tools/isa_seq.py file.s <mangled-name-substring>
M mfma, v VALU, r ds_read, w ds_write, G global/buffer load, S global store, | s_waitcnt, B s_barrier, J branch, s other scalar."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r"^(_Z\S*):\s*;[^\n]*\n(.*?)^\s*\.end_amdhsa_kernel", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    def grab(k):
        r = re.search(r"\.amdhsa_%s (\S+)" % k, body)
        return r.group(1) if r else None
    print(name)
    print("  next_free_vgpr", grab("next_free_vgpr"), "accum_offset", grab("accum_offset"), "scratch", grab("private_segment_fixed_size"), "lds", grab("group_segment_fixed_size"))
    lines = body.split("\n")
    # basic blocks
    blocks, cur, label = [], [], "entry"
    for l in lines:
        t = l.strip()
        if re.match(r"^\.LBB\S+:", t):
            blocks.append((label, cur)); cur, label = [], t.split(":")[0]
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t)
    blocks.append((label, cur))
    def cls(op):
        if op.startswith("v_mfma"): return "M"
        if op.startswith("ds_read") or op.startswith("ds_load"): return "r"
        if op.startswith("ds_write") or op.startswith("ds_store"): return "w"
        if op.startswith("global_load") or op.startswith("buffer_load"): return "G"
        if op.startswith("global_store") or op.startswith("buffer_store"): return "S"
        if op.startswith("v_"): return "v"
        if op.startswith("s_waitcnt"): return "|"
        if op.startswith("s_barrier"): return "B"
        if op.startswith("s_cbranch") or op.startswith("s_branch"): return "J"
        if op.startswith("s_"): return "s"
        return "?"
    for label, b in blocks:
        n = sum(1 for t in b if t.startswith("v_mfma"))
        if n:
            seq = "".join(cls(t.split()[0]) for t in b)
            print(f"  block {label}: {len(b)} instr, {n} mfma, {seq.count('v')} valu, {seq.count('r')} ds_read, {seq.count('w')} ds_write, {seq.count('G')} loads")
            print("   ", seq)
