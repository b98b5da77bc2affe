"""Reads the gfx950 code objects embedded in libmgld_hip.so (clang offload bundles in .hip_fatbin) and returns the AMDGPU metadata of every
kernel (register counts, spills, scratch) — test infrastructure, no GPU and no ROCm tool needed."""
import struct

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _bundles(blob):
    i = 0
    while True:
        i = blob.find(MAGIC, i)
        if i < 0:
            return
        n, = struct.unpack_from("<Q", blob, i + 24)
        o = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, o)
            triple = blob[o + 24:o + 24 + tl].decode()
            o += 24 + tl
            yield triple, blob[i + off:i + off + size]
        i += 24


def _metadata_notes(elf):
    assert elf[:4] == b"\x7fELF"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for k in range(shnum):
        sh = shoff + k * shentsize
        typ, = struct.unpack_from("<I", elf, sh + 4)
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        if typ != 7:          # SHT_NOTE
            continue
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if name.startswith(b"AMDGPU") and ntype == 32:      # NT_AMDGPU_METADATA (msgpack)
                yield msgpack.unpackb(desc, raw=False, strict_map_key=False)


def kernels(so_path, arch="gfx950"):
    """-> list of the `amdhsa.kernels` metadata dicts of every kernel compiled for `arch`"""
    blob = open(so_path, "rb").read()
    out = []
    for triple, code in _bundles(blob):
        if arch not in triple or not code:
            continue
        for md in _metadata_notes(code):
            out.extend(md.get("amdhsa.kernels", []))
    return out
