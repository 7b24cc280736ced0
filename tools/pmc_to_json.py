#!/usr/bin/env python
"""profiles/<tag>_pmc_traffic.json from the two PMC summaries tools/collect_round.sh leaves in gpurun_out/
   (tools/pmc_summary.py output of the FETCH_SIZE and the WRITE_SIZE pass):   python tools/pmc_to_json.py r02 <git hash>"""
import json, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
commit = sys.argv[2] if len(sys.argv) > 2 else "unknown"
SHORT = [("ray_encode", "ray_encode_pair"), ("slab_accumulate", "slab_accumulate"), ("slab_combine", "slab_combine"), ("scatter_fill", "scatter_fill"),
         ("shade_bwd", "shade_bwd"), ("shade_fwd", "shade_fwd"), ("wgrad_mlp_kernelILb0", "wgrad_mlp_sdf"),
         ("wgrad_mlp_kernelILb1", "wgrad_mlp_geo"), ("wgrad_mlp_kernel<false", "wgrad_mlp_sdf"), ("wgrad_mlp_kernel<true", "wgrad_mlp_geo"),
         ("wgrad_dec", "wgrad_dec"), ("wgrad_tail", "reduce_finalize"), ("wgrad_reduce_all", "wgrad_reduce_all"), ("post_shade", "post_shade"), ("finalize", "finalize")]


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"(.*?)\s+(\S+)\s+n=\s*(\d+)\s+mean=\s*([0-9.]+)", line)
        if not m or m.group(2) != counter:
            continue
        for key, name in SHORT:
            if key in m.group(1):
                out[name] = float(m.group(4)) * 1024.0          # counter unit: KB
                break
    return out


fetch = read(f"gpurun_out/{tag}_pmc_fetch.txt", "FETCH_SIZE")
write = read(f"gpurun_out/{tag}_pmc_write.txt", "WRITE_SIZE")
doc = {"_about": "HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes with --kernel-trace "
                 "only, LS2FM_SERIAL=1, bench.py default workload C2), counter KB * 1024.  `fetch_raw` is the RAW counter; `fetch` "
                 "applies the guide's gfx950 correction for wide coalesced streaming reads (FETCH_SIZE reports half: "
                 "MI355X_MICROARCH.md, HBM section) -- calibrated on this hardware for streaming reads AND for 8- / 16-byte gathers of scattered lines "
                 "(tools/fetch_calib.hip, profiles/r06_raw/c83_fetch_calib.txt: 1 GiB of distinct 128-byte lines touched -> FETCH_SIZE "
                 "512 MiB, one TCC miss and one EA read request per line, whatever part of the line is used; the gathers take as long "
                 "as streaming the lines whole).  Infinity-Cache hits are counted.  "
                 "bench.py quotes fetch + write as roofline.traffic.",
       "_commit": commit}
for name in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, 0) + write.get(k, 0))):
    doc[name] = {"fetch_raw": fetch.get(name, 0.0), "fetch": 2 * fetch.get(name, 0.0), "write": write.get(name, 0.0)}
if "scatter_fill" in doc and "slab_accumulate" in doc:
    # the table-gradient scatter AS LAUNCHED: both passes with everything that rides in them (the weight-gradient tail's jobs inside
    # the fill launch, the split slabs' combine inside the accumulate launch) -- what the pair's roofline.traffic is quoted from
    doc["scatter_pair_as_launched"] = {k: doc["scatter_fill"][k] + doc["slab_accumulate"][k] for k in ("fetch_raw", "fetch", "write")}
    doc["scatter_pair_as_launched"]["total"] = doc["scatter_pair_as_launched"]["fetch"] + doc["scatter_pair_as_launched"]["write"]
json.dump(doc, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if not k.startswith("_")}, indent=1)[:1500])
