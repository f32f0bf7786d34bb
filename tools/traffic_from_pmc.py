#!/usr/bin/env python3
"""profiles/traffic.json from the per-dispatch FETCH_SIZE / WRITE_SIZE files of tools/profile_round.sh:

    python tools/traffic_from_pmc.py gpurun_out/profiles_r02 r02 > profiles/traffic.json

Per workload: the median over the step kernel's dispatches of each counter (KiB); HBM bytes per launch =
(2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE doubled as MI355X_MICROARCH.md (section HBM) prescribes for gfx950: the
counter reports half the bytes of a wide coalesced read stream; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
Both counters sit on the L2's memory-side (fabric) interface: Infinity-Cache hits are counted, so below ~0.6 Mi envs the
"HBM" bytes include reads the 256 MiB cache served (profiles/README.md says which workloads fit it)."""
import csv
import glob
import json
import os
import sys

WORKLOADS = {  # name -> (traffic.json key, layout bytes per env-step, envs)
    "empty8x8_1M": ("MiniGrid-Empty-8x8-v0/partial/1048576", 233, 1048576),
    "doorkey8x8_1M": ("MiniGrid-DoorKey-8x8-v0/partial/1048576", 233, 1048576),
    "lavacrossing_512k": ("MiniGrid-LavaCrossingS9N1-v0/partial/524288", 253, 524288),
    "lavacrossing_1M": ("MiniGrid-LavaCrossingS9N1-v0/partial/1048576", 253, 1048576),
    "empty16x16_full_256k": ("MiniGrid-Empty-16x16-v0/full/262144", 1046, 262144),
    "empty8x8_4M": ("MiniGrid-Empty-8x8-v0/partial/4194304", 233, 4194304),
    "lavacrossing_4M": ("MiniGrid-LavaCrossingS9N1-v0/partial/4194304", 253, 4194304),
    # the gather form (k_step<0,0,3,7>) reads its 7 x 8-byte window + the forward cell, not the grid: 57 + 16 + 1 + 147 + 5 = 226 B
    "multiroom_n6_256k": ("MiniGrid-MultiRoom-N6-v0/partial/262144", 226, 262144),
    "fourrooms_1M": ("MiniGrid-FourRooms-v0/partial/1048576", 226, 1048576),
    "empty16x16_512k": ("MiniGrid-Empty-16x16-v0/partial/524288", 226, 524288),
    # Dynamic-Obstacles, walk fused into the step (k_step_dyn): tile read + written back, obstacle order, RNG position, tape window: 345 B
    "dynobs8x8_1M": ("MiniGrid-Dynamic-Obstacles-8x8-v0/partial/1048576", 345, 1048576),
}


def median_counter(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and "k_step" in r["Kernel_Name"]]
    vals.sort()
    return (vals[len(vals) // 2], len(vals)) if vals else (None, 0)


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (--kernel-trace only) of `python bench.py --steps 24 "
                   "--warmup 8 --no-cpu-baseline <workload>` (tools/profile_round.sh %s pmc; per-dispatch files under profiles/%s_pmc/); "
                   "median over the step kernel's dispatches; counters are KiB; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                   "(gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md); the counters sit on the L2's fabric side, so Infinity-Cache "
                   "hits are included (see profiles/README.md)." % (tag, tag)}
    for name, (key, bps, n) in WORKLOADS.items():
        f = os.path.join(d, "%s_pmc_%s_FETCH_SIZE.csv" % (tag, name))
        w = os.path.join(d, "%s_pmc_%s_WRITE_SIZE.csv" % (tag, name))
        if not (os.path.exists(f) and os.path.exists(w)):
            continue
        fv, fn = median_counter(f, "FETCH_SIZE")
        wv, wn = median_counter(w, "WRITE_SIZE")
        if fv is None or wv is None:
            continue
        out[key] = {"fetch_size_kib_raw": fv, "write_size_kib": wv, "dispatches": [fn, wn],
                    "hbm_bytes_per_launch": int((2 * fv + wv) * 1024), "expected_layout_bytes_per_launch": bps * n}
    json.dump(out, sys.stdout, indent=2)
    print()


if __name__ == "__main__":
    main()
