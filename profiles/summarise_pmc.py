"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into profiles/pmc_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python profiles/summarise_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write "<the bench command>" <steps run> \
        [<batch per GPU> [<output json, default profiles/pmc_traffic.json>]]

Both counters are in KiB.  Correction applied (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE tallies the
128-byte read requests at 64 bytes, so reads = 2 * FETCH_SIZE; WRITE_SIZE is exact.  The calibration rows of the
output check that on kernels of this build with a known byte count (bn_bwd_apply: reads 2 tensors, writes 1).
"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(directory):
    rows = collections.defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows


def family(name):
    if "convbf_res_kernel" in name:          # the resident-weights form is reported under its family (avsep_conv_kernel_name: convbf_kernel)
        return "convbf_kernel"
    for key in ("winow4_kernel", "wino4_kernel", "winow_kernel", "wino_kernel", "convbf_kernel", "wgradb_reduce_kernel", "wgradb_kernel", "wgradbf_kernel", "wgrad4bf_kernel", "wgrad4d_kernel", "smallci_wgrad_kernel",
                "conv3x3_kernel", "wgrad3x3_kernel", "head_fwd_kernel", "head_wgrad_kernel", "affine_act_kernel", "maxpool_fwd4_kernel",
                "head_dgrad_kernel", "smallco_fwd", "smallco_wgrad", "smallci_dgrad", "relu_up2x_fwd",
                "relu_up2x_bwd", "bn_bwd_apply_kernel", "affine_act_bwd_kernel", "sgd_kernel", "w3_reduce_kernel"):
        if key in name:
            return key
    if "igemm_kernel<" in name:
        return "igemm_kernel<%s>" % {"0": "fwd", "1": "dgrad", "2": "wgrad"}[name.split("igemm_kernel<")[1][0]]
    # every other kernel of the step under its own base name (template arguments and parameter list stripped), so that the
    # per-step total covers the WHOLE step (round 3's total only summed the families listed above)
    base = name.replace("void ", "").split("<")[0].split("(")[0].strip()
    return base or None


def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    fams = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for name, vals in fetch.items():
        k = family(name)
        if k:
            fams[k][0] += len(vals); fams[k][1] += sum(vals)
    for name, vals in write.items():
        k = family(name)
        if k:
            fams[k][2] += len(vals); fams[k][3] += sum(vals)
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # train steps in each pass (warm-up + timed)
    out = {"command": sys.argv[3] if len(sys.argv) > 3 else "", "steps_per_pass": steps,
           "batch": int(sys.argv[5]) if len(sys.argv) > 5 else None,
           "unit": "bytes per kernel launch (average over the family's launches); *_per_step = family total / steps",
           "correction": "reads = 2 * FETCH_SIZE KiB (gfx950), writes = WRITE_SIZE KiB", "kernels": {}}
    for k, (nf, sf, nw, sw) in sorted(fams.items()):
        if nf == 0 or nw == 0:
            continue
        rd, wr = 2.0 * sf / nf * 1024.0, sw / nw * 1024.0
        out["kernels"][k] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_size_kib_raw": sf / nf,
                             "write_size_kib": sw / nw, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                             "traffic_bytes_per_launch": rd + wr,
                             "traffic_bytes_per_step": (rd + wr) * nf / steps if steps else None}
    if steps:
        out["total_traffic_bytes_per_step"] = sum(v["traffic_bytes_per_step"] for v in out["kernels"].values())
        print("total_traffic_GB_per_step", out["total_traffic_bytes_per_step"] / 1e9)
    dst = sys.argv[6] if len(sys.argv) > 6 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k in ("wino4_kernel", "winow4_kernel", "wino_kernel", "winow_kernel", "conv3x3_kernel", "convbf_kernel", "wgradb_kernel"):
        if k in out["kernels"]:
            print(k, json.dumps(out["kernels"][k]))


if __name__ == "__main__":
    main()
