"""Registers / LDS / scratch of every kernel in the built library, read from the code objects inside the .so (the
NT_AMDGPU_METADATA note of each gfx950 ELF in its clang offload bundles) - no compiler, no GPU.
    python tools/kernel_resources.py [pattern]
`waves_per_simd(k)` applies what the hardware grants on gfx950 (512 VGPRs and 800 SGPRs per SIMD lane group, SGPRs
allotted in 16s plus 16 per wave for the trap handler, at most 8 waves): the compiler's own "Occupancy" remark ignores
the trap handler's share, and `raster_fwd_kernel` fell from two blocks per CU to one at 83 SGPRs."""
import os
import re
import struct
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "indirect_learning_pose-shape_amd", "libsmplraster_hip.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    for m in re.finditer(MAGIC, blob):
        o = m.start()
        (n,) = struct.unpack_from("<Q", blob, o + len(MAGIC))
        p = o + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl].decode()
            p += tl
            if size and "gfx950" in triple:
                yield blob[o + off:o + off + size]


def _metadata(elf):
    """The msgpack map of the NT_AMDGPU_METADATA (type 32, name "AMDGPU") note of a 64-bit little-endian ELF."""
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        if sh_type != 7:                                   # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                return msgpack.unpackb(desc, raw=False, strict_map_key=False)
    return None


def kernels(lib=LIB):
    """{kernel name: {sgpr, vgpr, agpr, lds, scratch, max_threads}} over every code object of the library."""
    out = {}
    with open(lib, "rb") as f:
        blob = f.read()
    for elf in _code_objects(blob):
        md = _metadata(elf)
        for k in (md or {}).get("amdhsa.kernels", []):
            out[k[".name"]] = {"sgpr": k[".sgpr_count"], "vgpr": k[".vgpr_count"], "agpr": k.get(".agpr_count", 0),
                               "lds": k[".group_segment_fixed_size"], "scratch": k[".private_segment_fixed_size"],
                               "max_threads": k[".max_flat_workgroup_size"]}
    return out


def waves_per_simd(k):
    by_sgpr = 800 // (((k["sgpr"] + 15) & ~15) + 16)
    regs = ((k["vgpr"] + k["agpr"] + 7) & ~7) or 8          # unified register file, allotted in 8s
    return max(1, min(8, by_sgpr, 512 // regs))


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else "."
    for name, k in sorted(kernels().items()):
        if re.search(pat, name):
            print("%-86s sgpr %3d vgpr %3d agpr %3d lds %6d scratch %4d -> %d waves / SIMD"
                  % (name[:86], k["sgpr"], k["vgpr"], k["agpr"], k["lds"], k["scratch"], waves_per_simd(k)))
