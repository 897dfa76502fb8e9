#!/usr/bin/env python3
"""Diagnostic: per-basic-block instruction mix of one kernel in a hipcc -S listing.
usage: isa_blocks.py listing.s kernel_symbol_substring [min_loads]"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]; minl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("ZlBatch") or (l.startswith("_Z") and key in l))
end = next(i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in txt[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((";", "//")): continue
    if t.endswith(":") or re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append((name, cur)); cur = []; name = t.split(":")[0]; continue
    if t.startswith("."): continue
    cur.append(t.split()[0])
blocks.append((name, cur))
for name, ins in blocks:
    c = Counter(ins)
    loads = sum(v for k, v in c.items() if k.startswith("global_load"))
    if loads < minl: continue
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
    f64 = sum(v for k, v in c.items() if "f64" in k)
    ds = sum(v for k, v in c.items() if k.startswith("ds_"))
    sal = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith("s_waitcnt"))
    print(f"{name:14s} n={len(ins):4d} valu={valu:4d} (pk {pk}, f64 {f64}) ds={ds:3d} salu={sal:3d} gloads={loads} waitcnt={c.get('s_waitcnt',0)}")
