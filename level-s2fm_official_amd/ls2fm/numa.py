"""Process placement: run on the CPUs of the GPU's NUMA node.

Measured on MI355X (profiles/r06_notes.md section 6): the host memory the HIP runtime sets up when a process initialises -- the AQL
queues among it -- lands on the NUMA node of the thread that touches it first, and whatever the GPU reads from it pays that node's
distance: `shade_bwd`, whose waves read the dispatch packet until round 6, ran 123 us from the GPU's node and 135 us from the other
socket's.  The kernels no longer make such reads (tools/check_host_reads.py keeps it that way), so nothing measured depends on this
module any more; one process per GPU, pinned to the GPU's node BEFORE the runtime initialises, is still what a launcher's
`numactl --cpunodebind` does (host-side launch latency, pinned staging buffers), and this module does it from inside the process,
without touching the HIP runtime: the GPU's PCI address comes from the KFD topology in sysfs.

    from ls2fm.numa import bind_to_gpu_numa_node
    bind_to_gpu_numa_node(local_rank)          # before the first torch.cuda / ls2fm call;  LS2FM_NUMA_BIND=0 turns it off
"""
from __future__ import annotations

import glob
import os
from typing import Optional, Set


def _parse_cpulist(text: str) -> Set[int]:
    cpus: Set[int] = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _visible_index(device_index: int) -> int:
    """position in the KFD topology's GPU list of the `device_index`-th VISIBLE device (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
    hold indices there; UUID forms are not resolved: the caller then gets None)"""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            ids = [t.strip() for t in v.split(",") if t.strip()]
            return int(ids[device_index])           # ValueError for UUIDs -> caught by the caller
    return device_index


def gpu_numa_node(device_index: int = 0) -> Optional[int]:
    """NUMA node of the `device_index`-th GPU (KFD topology order = HIP's enumeration order), or None when sysfs does not say"""
    try:
        gpus = []
        for node in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*"), key=lambda p: int(os.path.basename(p))):
            try:        # (a container sees every node of the box but may read only its own GPUs' properties: those are HIP's devices)
                props = dict(line.split() for line in open(os.path.join(node, "properties")) if len(line.split()) == 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        p = gpus[_visible_index(device_index)]
        loc, dom = int(p["location_id"]), int(p.get("domain", "0"))
        addr = f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7:x}"
        node = int(open(f"/sys/bus/pci/devices/{addr}/numa_node").read().strip())
        return node if node >= 0 else None
    except (OSError, ValueError, KeyError, IndexError):
        return None


def bind_to_gpu_numa_node(device_index: int = 0) -> Optional[int]:
    """restrict the calling thread (and every thread it starts from now on) to the CPUs of the GPU's NUMA node -- intersected with the
    CPUs the process may use already.  Returns the node, or None when nothing was changed (switch off, unknown topology, no CPU left)."""
    if os.environ.get("LS2FM_NUMA_BIND", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    node = gpu_numa_node(device_index)
    if node is None:
        return None
    try:
        cpus = _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) & os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except (OSError, ValueError):
        return None
