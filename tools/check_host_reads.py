#!/usr/bin/env python
"""Which kernels of libls2fm_hip.so ask for the DISPATCH or QUEUE pointer (kernel descriptor bits 1 / 2 of kernel_code_properties)?
Both point into the AQL queue, which lives in HOST memory: a kernel that reads through them (blockDim / gridDim taken from the dispatch
packet; a private array promoted to LDS and indexed by the flat work-item id) makes uncached reads over PCIe from its waves -- round 6
found `shade_bwd` doing that ~290 times per launch, which made it 123 us from the GPU's NUMA node and 135 us from the other socket.
No GPU, no ROCm tool: parses the clang offload bundle inside the .so and the code objects' ELF symbol tables.
    python tools/check_host_reads.py [path/to/lib.so]   -> lists offenders, exit status 1 if any"""
import os, struct, sys


def code_objects(blob):
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = 0
    while True:
        at = blob.find(magic, at)
        if at < 0:
            return
        n, = struct.unpack_from("<Q", blob, at + 24)
        p = at + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                yield triple, blob[at + off:at + off + size]
        at += 24


def kernel_descriptors(elf):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    for name, typ, flags, addr, off, size, link, info, align, entsize in secs:
        if typ not in (2, 11):                     # SHT_SYMTAB, SHT_DYNSYM
            continue
        stroff = secs[link][4]
        for i in range(size // 24):
            st_name, st_info, st_other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, off + 24 * i)
            end = elf.index(b"\0", stroff + st_name)
            sym = elf[stroff + st_name:end].decode()
            if sym.endswith(".kd") and 0 < st_shndx < shnum:
                s = secs[st_shndx]
                kd = elf[s[4] + st_value - s[3]:s[4] + st_value - s[3] + 64]
                props, = struct.unpack_from("<H", kd, 56)
                yield sym[:-3], props
        if typ == 2:
            return


def offenders(path):
    out = []
    seen = set()
    for triple, elf in code_objects(open(path, "rb").read()):
        for name, props in kernel_descriptors(elf):
            if name in seen:
                continue
            seen.add(name)
            if props & 0b110:
                out.append((name, "dispatch_ptr" if props & 2 else "", "queue_ptr" if props & 4 else ""))
    return out, len(seen)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "level-s2fm_official_amd", "ls2fm", "libls2fm_hip.so")
    bad, n = offenders(lib)
    for name, d, q in bad:
        print(f"{name[:110]}  {d} {q}")
    print(f"{n} kernels, {len(bad)} read through the dispatch / queue pointer")
    sys.exit(1 if bad else 0)
