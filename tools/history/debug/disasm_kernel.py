"""Disassemble the kernels of libguardx_hip.so whose mangled name contains every given substring:
    python tools/debug/disasm_kernel.py dyn_tape_kernel PointRobot Lb1E > /tmp/dyn.s"""
import os, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
import kernel_descriptors as kd
from guardx_amd import _native
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
with tempfile.TemporaryDirectory() as tmp:
    for i, (triple, data) in enumerate(kd.code_objects(_native.LIB_PATH)):
        if "gfx950" not in triple or data[:4] != b"\x7fELF":
            continue
        names = [k for k in kd.descriptors(data) if all(s in k for s in sys.argv[1:])]
        if not names:
            continue
        path = os.path.join(tmp, f"co{i}.elf")
        open(path, "wb").write(data)
        for n in names:
            print("//", n)
            print(subprocess.run([objdump, "-d", "--mcpu=gfx950", "--disassemble-symbols=" + n, path],
                                 capture_output=True, text=True, check=True).stdout)
