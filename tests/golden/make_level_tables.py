#!/usr/bin/env python
"""Per-level geometry of the shipped hash-grid configuration (options/config_hash_sdf.json: L16 / F2 / T2^19 / N_min 16) for
the four dataset presets, written out as LITERAL values -> tests/golden/level_tables.json.

The per-level scale comes from the REFERENCE (models/base.py:120-139, imported); the level table itself is tcnn's
(third-party, not in the reference tree): it is produced here by the oracle's restatement of tcnn 1.7 `grid.h` and therefore
unpinned -- which is exactly why it is written down: a future cross-check against a real tinycudann build compares THESE numbers.
For every level the file also records which resolutions `ceil(exp2f(l * log2f(b)) * 16 - 1) + 1` can take when log2f is off by
up to 1 ulp and exp2f by up to 2 (CUDA's documented bounds): a level with more than one is where a real-tcnn checkpoint may be
indexed differently from this build (which DEFINES the scale with correctly rounded log2 / exp2).
Build container only (imports /root/reference for the per-level scale)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    import make_golden as MG
    MG.install_stubs()
    sys.path.insert(0, MG.REF)
    os.chdir(MG.REF)
    import warnings
    warnings.filterwarnings("ignore")
    from models import base as ref_base
    out = {}
    for ds in ("DTU", "ETH3D", "BlendedMVS", "scannet"):
        opt = MG.make_opt(ds, os.path.join(MG.REF, "options/config_hash_sdf.json"), False, 128)
        t = ref_base.get_Embedder(opt=opt, input_dim=3, input_choice="Hash").embedder_obj.table
        b = np.float32(t.per_level_scale)
        log2b = np.float32(np.log2(np.float64(b)))
        levels = []
        def step(x, k):
            for _ in range(abs(k)):
                x = np.nextafter(x, np.float32(np.inf if k > 0 else -np.inf))
            return x
        for l in range(t.n_levels):
            # the envelope of a device libm (CUDA: log2f <= 1 ulp, exp2f <= 2 ulp); level 0 is exact in any libm (exp2f(0) = 1)
            seen = set()
            for dl in ((0,) if l == 0 else (-1, 0, 1)):
                for de in ((0,) if l == 0 else (-2, -1, 0, 1, 2)):
                    e = step(np.float32(np.exp2(np.float64(np.float32(l) * step(log2b, dl)))), de)
                    seen.add(int(np.ceil(np.float32(e * np.float32(16) - np.float32(1.0)))) + 1)
            levels.append(dict(level=l, scale=float(t.scale[l]), resolution=int(t.resolution[l]), size=int(t.size[l]),
                               offset=int(t.offset[l]), hashed=bool(t.hashed[l]),
                               resolutions_within_libm_envelope=sorted(seen), ulp_sensitive=len(seen) > 1))
        out[ds] = dict(per_level_scale=float(t.per_level_scale), n_params=int(t.n_params), levels=levels)
        flagged = [lv["level"] for lv in levels if lv["ulp_sensitive"]]
        print(f"[levels] {ds}: b={t.per_level_scale:.6f} params={t.n_params} top res {levels[-1]['resolution']} ulp-sensitive levels {flagged}")
    with open(os.path.join(HERE, "level_tables.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
