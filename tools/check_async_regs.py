#!/usr/bin/env python
"""`slab_accumulate_persistent_kernel` issues scalar loads and its unit claim (a returning global_atomic_add) BY HAND and completes
them later (`s_waitcnt lgkmcnt(0)` in front of the barrier behind the flush; `s_waitcnt vmcnt(0)` behind the unit's adds): the compiler
does not know that their destination registers are in flight.  This walks the kernel's ISA in the shipped .so and follows EVERY path
from each such instruction to the wait that completes it; an instruction on the way that reads or writes the destination register (a
copy, a spill, a re-use) is an error.  Checked for every scalar load of those kernels (the compiler's own satisfy it by construction)
and for every returning atomic issued under `s_mov_b64 exec, 1`.  No GPU; needs llvm-objdump of the ROCm install.
    python tools/check_async_regs.py [path/to/lib.so]   -> exit status 1 on a violation"""
import os, re, subprocess, sys, tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_host_reads import code_objects                                   # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
KERNELS = ("slab_accumulate_persistent_kernel",)


def regs(text):
    """(kind, index) of every register an operand string names: s8, s[6:7], v73, v[10:13], a3, vcc / exec do not matter here."""
    out = set()
    for k, lo, hi in re.findall(r"\b([sva])\[(\d+):(\d+)\]", text):
        out.update((k, i) for i in range(int(lo), int(hi) + 1))
    for k, i in re.findall(r"\b([sva])(\d+)\b", text):
        out.add((k, int(i)))
    return out


def functions(dis):
    cur, rows = None, []
    for line in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            if cur:
                yield cur, rows
            cur, rows = m.group(2), []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur:
            rows.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if cur:
        yield cur, rows


def successors(rows, at, i):
    addr, op, args = rows[i]
    nxt = []
    if op.startswith(("s_branch", "s_cbranch")):
        off = int(args.split()[0])                                          # the 16-bit field, printed unsigned
        target = addr + 4 + 4 * (off - 65536 if off >= 32768 else off)
        if target in at:
            nxt.append(at[target])
        if op.startswith("s_branch"):
            return nxt
    if op in ("s_endpgm", "s_setpc_b64"):
        return nxt
    if i + 1 < len(rows):
        nxt.append(i + 1)
    return nxt


def walk(rows, at, start, dest, done):
    """every instruction reachable from `start` before a wait for which done(args) holds; returns those that touch `dest`"""
    bad, seen, todo = [], set(), list(successors(rows, at, start))
    while todo:
        i = todo.pop()
        if i in seen:
            continue
        seen.add(i)
        addr, op, args = rows[i]
        if op == "s_waitcnt" and done(args):
            continue
        if i != start and regs(args) & dest:
            bad.append((addr, op, args))
        todo.extend(successors(rows, at, i))
    return bad


def check(path):
    blob = open(path, "rb").read()
    n_sload = n_atomic = 0
    errors = []
    for _, elf in code_objects(blob):
        if not any(k.encode() in elf for k in KERNELS):
            continue
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(elf)
            f.flush()
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True).stdout
        for name, rows in functions(dis):
            if not any(k in name for k in KERNELS):
                continue
            at = {a: i for i, (a, _, _) in enumerate(rows)}
            for i, (addr, op, args) in enumerate(rows):
                if op.startswith("s_load_dword") or op.startswith("s_buffer_load"):
                    n_sload += 1
                    dest = regs(args.split(",")[0])
                    bad = walk(rows, at, i, dest, lambda a: "lgkmcnt(0)" in a)
                elif op.startswith("global_atomic") and " sc0" in " " + args and any(
                        r[1] == "s_mov_b64" and r[2].replace(" ", "") == "exec,1" for r in rows[max(0, i - 3):i]):
                    n_atomic += 1
                    dest = regs(args.split(",")[0])
                    bad = walk(rows, at, i, dest, lambda a: "vmcnt(0)" in a)
                else:
                    continue
                for b in bad:
                    errors.append(f"{name[:70]} +{addr:#x} {op} {args.split(',')[0]}: touched in flight at {b[0]:#x}: {b[1]} {b[2]}")
    return errors, n_sload, n_atomic


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "level-s2fm_official_amd", "ls2fm", "libls2fm_hip.so")
    errors, n_sload, n_atomic = check(lib)
    for e in errors:
        print(e)
    print(f"{n_sload} scalar loads and {n_atomic} hand-issued returning atomics followed to their waits, {len(errors)} touched in flight")
    sys.exit(1 if errors else 0)
