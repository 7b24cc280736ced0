#!/usr/bin/env python
"""VGPRs / scratch / LDS / occupancy of every kernel of the library as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).  A kernel with a non-zero ScratchSize is a red flag: the gather
pass once lost 8x to a 652-byte-per-lane scratch copy of its level tables.   python tools/kernel_resources.py [file.hip ...]"""
import glob, os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "level-s2fm_official_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(root, "*.hip")))
flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage".split()
per_file = {"render_fwd.hip": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]}      # = the Makefile's FLAGS_<file>
bad = 0
for f in files:
    out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, *per_file.get(os.path.basename(f), []), "-c", f, "-o", "/dev/null"], capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    for k, v in rows.items():
        flag = "  <-- SCRATCH" if v.get("ScratchSize", 0) else ""
        bad += bool(flag)
        print(f"{os.path.basename(f):18s} {k[:58]:58s} vgpr {v.get('VGPRs', -1):4d} agpr {v.get('AGPRs', 0):4d} sgpr {v.get('TotalSGPRs', -1):4d} "
              f"scratch {v.get('ScratchSize', 0):5d} lds {v.get('LDS Size', 0):7d} occ {v.get('Occupancy', -1)}{flag}")
sys.exit(1 if bad else 0)
