#!/usr/bin/env python3
"""Disassemble one gfx950 kernel of libguardx_hip.so and print its instruction mix (whole kernel, or the largest loop):
    python tools/disasm_kernel.py <substring of the mangled name> [--lib path] [--dump out.s]"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_descriptors as kd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("pattern")
ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "guardx_amd", "lib",
                                              "libguardx_hip.so"))
ap.add_argument("--dump")
ap.add_argument("--all", action="store_true", help="every match (default: the first)")
ap.add_argument("--loops", action="store_true", help="also list every loop (backward branch) of 100+ instructions with its size and VALU count")
args = ap.parse_args()
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
pats = args.pattern.split("&")
with tempfile.TemporaryDirectory() as tmp:
    for i, (triple, data) in enumerate(kd.code_objects(args.lib)):
        if "gfx950" not in triple or data[:4] != b"\x7fELF":
            continue
        names = [k for k in kd.descriptors(data) if all(p in k for p in pats)]
        if not names:
            continue
        path = os.path.join(tmp, f"co{i}.elf")
        open(path, "wb").write(data)
        for n in names if args.all else names[:1]:
            asm = subprocess.run([objdump, "-d", "--mcpu=gfx950", "--disassemble-symbols=" + n, path], capture_output=True,
                                 text=True, check=True).stdout
            if args.dump:
                open(args.dump, "w").write(asm)
            ins = []
            for ln in asm.splitlines():
                m = re.match(r"^\s+([a-z_0-9]+)\s.*//\s*([0-9A-F]+):", ln)
                if m:
                    ins.append((int(m.group(2), 16), m.group(1), ln))
            # largest backward branch = the step loop
            best = None
            for a, op, ln in ins:
                if op.startswith("s_cbranch") or op == "s_branch":
                    m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", ln)
                    if m:
                        # target offset relative to the symbol start
                        tgt = ins[0][0] + int(m.group(1), 16)
                        if tgt < a and (best is None or a - tgt > best[1] - best[0]):
                            best = (tgt, a)
            print(n[:150])
            if args.loops:
                seen = set()
                for a, op, ln in ins:
                    if op.startswith("s_cbranch") or op == "s_branch":
                        m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", ln)
                        if m:
                            tgt = ins[0][0] + int(m.group(1), 16)
                            body = [x for x in ins if tgt <= x[0] <= a]
                            if tgt < a and len(body) >= 100 and (tgt) not in seen:
                                seen.add(tgt)
                                nv = sum(1 for x in body if x[1].startswith("v_"))
                                nd = sum(1 for x in body if "dpp" in x[2] or "row_" in x[2] or "quad_perm" in x[2])
                                print(f"  loop of {len(body)} instructions: {nv} VALU, of them {nd} DPP, "
                                      f"{sum(1 for x in body if x[1].startswith('v_rcp'))} divisions")
            for label, sel in (("kernel", ins), ("largest loop", [x for x in ins if best and best[0] <= x[0] <= best[1]])):
                c = collections.Counter()
                for a, op, ln in sel:
                    key = ("dpp " if "dpp" in ln or "row_" in ln or "quad_perm" in ln else "") + \
                          ("v_pk" if op.startswith("v_pk") else op.split("_e")[0] if op.startswith("v_") else
                           "s_nop" if op == "s_nop" else "s_waitcnt" if op == "s_waitcnt" else "salu" if op.startswith("s_") else
                           "ds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else op)
                    c[key] += 1
                valu = sum(v for k, v in c.items() if k.startswith(("v_", "dpp")))
                print(f"  {label}: {len(sel)} instructions, {valu} VALU")
                print("   ", ", ".join(f"{k} {v}" for k, v in c.most_common(28)))
