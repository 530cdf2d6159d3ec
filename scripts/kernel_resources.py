#!/usr/bin/env python3
"""Registers, LDS and scratch of every gfx950 kernel in a built library, read from the code objects' own metadata (no GPU, no tools):
the clang offload bundles inside the .so -> the amdgcn ELF of each translation unit -> its NT_AMDGPU_METADATA note (msgpack).

    python scripts/kernel_resources.py [cup3d_amd/libcup3d_hip.so] [--match loop] [--json]

Waves per SIMD follows from the unified register file of CDNA3/4: 512 VGPRs per lane and SIMD, allocated in blocks of 8, at most 8
wavefronts (arch + accumulation registers count together); LDS bounds the workgroups per CU (160 KiB on gfx950).
tests/test_kernel_resources.py pins the numbers DESIGN.md quotes for the production kernels."""
import argparse
import json
import os
import re
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_objects(path):
    """every gfx950 ELF bundled in the library: one per translation unit"""
    blob = open(path, "rb").read()
    out = []
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                out.append((triple, blob[base + off:base + off + size]))
    return out


def metadata(elf):
    """the msgpack document of the NT_AMDGPU_METADATA note (type 32, owner "AMDGPU") of a 64-bit little-endian ELF"""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2 and elf[5] == 1
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:  # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            d0 = p + 12 + (namesz + 3) // 4 * 4
            if ntype == 32 and name == b"AMDGPU":
                return msgpack.unpackb(elf[d0:d0 + descsz], raw=False, strict_map_key=False)
            p = d0 + (descsz + 3) // 4 * 4
    return None


def waves_per_simd(vgpr, agpr):
    regs = (max(1, vgpr + agpr) + 7) // 8 * 8
    return max(1, min(8, 512 // regs))


def demangle_short(name):
    """_ZN5cup3d10k_loop1_cgILb1ELi0ELb1EEEv... -> k_loop1_cg<b1,i0,b1> without calling c++filt (the names here are simple)"""
    m = re.match(r"_ZN?", name)
    if not m:
        return name
    p, base = m.end(), None
    while True:  # nested name: <length><identifier> components, the last one is the kernel
        c = re.match(r"(\d+)", name[p:])
        if not c:
            break
        n = int(c.group(1))
        base = name[p + c.end():p + c.end() + n]
        p += c.end() + n
    if base is None:
        return name
    t = re.match(r"I((?:L[bij]\d+E)+)E", name[p:])
    if t:
        base += "<" + ",".join(re.findall(r"L([bij]\d+)E", t.group(1))) + ">"
    return base


def kernels(path):
    rows = []
    for _, elf in code_objects(path):
        md = metadata(elf)
        for k in (md or {}).get("amdhsa.kernels", []):
            v, a = k.get(".vgpr_count", 0), k.get(".agpr_count", 0)
            rows.append({"kernel": demangle_short(k[".name"]), "symbol": k[".name"], "vgpr": v, "agpr": a, "sgpr": k.get(".sgpr_count", 0),
                         "lds_bytes": k.get(".group_segment_fixed_size", 0), "scratch_bytes": k.get(".private_segment_fixed_size", 0),
                         "vgpr_spills": k.get(".vgpr_spill_count", 0), "sgpr_spills": k.get(".sgpr_spill_count", 0),
                         "max_workgroup": k.get(".max_flat_workgroup_size", 0), "waves_per_simd": waves_per_simd(v, a)})
    return sorted(rows, key=lambda r: r["kernel"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("library", nargs="?", default=os.path.join(ROOT, "cup3d_amd", "libcup3d_hip.so"))
    ap.add_argument("--match", default="", help="only kernels whose name contains this")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    rows = [r for r in kernels(a.library) if a.match in r["kernel"]]
    if a.json:
        json.dump(rows, sys.stdout, indent=1)
        return
    print(f"# {os.path.relpath(a.library, ROOT)}: {len(rows)} kernels (registers / LDS / scratch from the code objects' metadata)")
    print(f"{'kernel':58s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'LDS B':>6s} {'scratch':>7s} {'spills':>6s} {'waves/SIMD':>10s}")
    for r in rows:
        print(f"{r['kernel'][:58]:58s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['lds_bytes']:6d} {r['scratch_bytes']:7d} {r['vgpr_spills']:6d} {r['waves_per_simd']:10d}")


if __name__ == "__main__":
    main()
