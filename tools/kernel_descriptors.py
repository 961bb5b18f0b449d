#!/usr/bin/env python3
"""Read the AMDGPU kernel descriptors out of libguardx_hip.so (no ROCm tools needed): name, LDS, scratch and the
user-SGPR enables.  A kernel with ENABLE_SGPR_DISPATCH_PTR / QUEUE_PTR reads the AQL packet -- host memory -- with a
scalar load at run time (~12 us per launch when it sits on the critical path).

    python tools/kernel_descriptors.py [path/to/lib.so]
"""
import os
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """(triple, bytes) of every entry of every offload bundle embedded in the shared object"""
    blob = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return out
        n, = struct.unpack_from("<Q", blob, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if size:
                out.append((triple, blob[i + off:i + off + size]))
        pos = i + 24


def descriptors(elf):
    """{kernel name: dict} from the .kd symbols of one AMDGPU code object (ELF64 little endian)"""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + k * shentsize) for k in range(shnum)]
    res = {}
    for s in secs:
        if s[1] not in (2, 11):          # SHT_SYMTAB / SHT_DYNSYM
            continue
        strtab = secs[s[6]]
        for k in range(s[5] // 24):
            name_off, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", elf, s[4] + 24 * k)
            end = elf.index(b"\0", strtab[4] + name_off)
            name = elf[strtab[4] + name_off:end].decode()
            if not name.endswith(".kd") or size != 64 or shndx == 0 or shndx >= shnum:
                continue
            sec = secs[shndx]
            kd = elf[sec[4] + value - sec[3]:sec[4] + value - sec[3] + 64]
            lds, scratch = struct.unpack_from("<II", kd, 0)
            props, = struct.unpack_from("<H", kd, 56)
            rsrc3, rsrc1 = struct.unpack_from("<II", kd, 44)
            res[name[:-3]] = dict(lds=lds, scratch=scratch, vgprs=((rsrc1 & 63) + 1) * 8,   # unified VGPR+AGPR file, granule 8
                                  accum_offset=((rsrc3 & 63) + 1) * 4, dispatch_ptr=bool(props & 2), queue_ptr=bool(props & 4),
                                  kernarg_ptr=bool(props & 8), dispatch_id=bool(props & 16))
    return res


def library_kernels(path):
    out = {}
    for triple, data in code_objects(path):
        if "gfx950" in triple and data[:4] == b"\x7fELF":
            out.update(descriptors(data))
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "guardx_amd", "lib", "libguardx_hip.so")
    ks = library_kernels(lib)
    bad = {k: v for k, v in ks.items() if v["dispatch_ptr"] or v["queue_ptr"]}
    print(len(ks), "kernels;", len(bad), "read the AQL dispatch/queue packet")
    for k in bad:
        print("  ", k)
    if len(sys.argv) > 2:                 # python tools/kernel_descriptors.py <lib> <substring>: resources of the matches
        import subprocess
        for k, v in sorted(ks.items()):
            if sys.argv[2] in k:
                try:
                    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", k], capture_output=True, text=True).stdout.strip()
                except OSError:
                    dem = k
                print(f"vgprs {v['vgprs']:4d} (agpr from {v['accum_offset']:3d})  lds {v['lds']:6d}  scratch {v['scratch']:5d}  {dem[:150]}")
