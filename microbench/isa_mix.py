#!/usr/bin/env python3
"""Instruction mix per kernel from `hipcc -S --cuda-device-only` output: python microbench/isa_mix.py file.s [filter]"""
import collections, re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\S+):\s", txt, flags=re.M)]
for (pos, name), nxt in zip(starts, starts[1:] + [(len(txt), None)]):
    body = txt[pos:nxt[0]].split(".amdhsa_kernel")[0].split(".end_amdhsa_kernel")[0]      # kernels with an early exit hold several s_endpgm
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::|\(.*$|^void ", "", dem)
    if flt not in dem:
        continue
    ops = collections.Counter(l.split()[0] for l in body.splitlines()
                              if l.startswith("\t") and l.split() and not l.strip().startswith((".", ";")))
    cls = lambda p: sum(v for k, v in ops.items() if k.startswith(p))
    print("%-46s total %5d valu %5d (pk %4d f64 %4d cvt %3d) salu %4d ds %3d vmem %3d lds_dma %2d" % (
        dem[:46], sum(ops.values()), cls("v_"), cls("v_pk_"), sum(v for k, v in ops.items() if k.startswith("v_") and "f64" in k),
        cls("v_cvt"), cls("s_"), cls("ds_"), cls("global_") + cls("buffer_") + cls("flat_"),
        sum(v for k, v in ops.items() if "lds" in k and k.startswith("global_load"))))
