#!/usr/bin/env python3
"""Distil two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of the SAME bench command,
with --kernel-trace only -- the gpurun rule) into profiles/rNN_pmc_traffic.json, the artefact bench.py reads for
`roofline.traffic`.

Per kernel: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024, averaged over the kernel's dispatches.
The factor 2 is the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per 128-B fabric
read request of a wide coalesced stream.  PMC records come per XCD / shader engine: they are summed per dispatch.

usage: python tools/pmc_traffic.py <fetch.db|fetch.csv> <write.db|write.csv> <per_gpu_batch> <run label> [out.json]"""
import json
import os
import sqlite3
import sys


def per_kernel(dbpath, counter):
    db = sqlite3.connect(dbpath)
    q = ("select s.kernel_name, count(distinct d.id), sum(e.value) from rocpd_pmc_event e "
         "join rocpd_info_pmc p on e.pmc_id = p.id "
         "join rocpd_kernel_dispatch d on d.event_id = e.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id where p.name = ? group by 1")
    return {name.split("(")[0]: (n, tot) for name, n, tot in db.execute(q, (counter,)).fetchall()}


def per_kernel_csv(path):
    """the same table from a tools/rocpd_pmc.py summary (kernel,counter,dispatches,sum,...) of the pass"""
    import csv
    return {r["kernel"].split("(")[0]: (int(r["dispatches"]), float(r["sum"])) for r in csv.DictReader(open(path))}


def main():
    if sys.argv[1].endswith(".csv"):
        fetch, write = per_kernel_csv(sys.argv[1]), per_kernel_csv(sys.argv[2])
    else:
        fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"per_gpu_batch": int(sys.argv[3]), "run": sys.argv[4], "formula": "(2*FETCH_SIZE + WRITE_SIZE) KB * 1024 per launch",
           "kernels": {}}
    short = {"conv3x3_halo_kernel": "conv3x3_halo_kernel", "conv_fwd_bx3_kernel": "conv_fwd_bx3_kernel",
             "linear_stream_bx3_kernel": "linear_stream_bx3_kernel", "attn_fwd2_bx3_kernel": "attn_fwd2_bx3_kernel",
             "attn_bwd_dkv_bx3_kernel": "attn_bwd_dkv_bx3_kernel", "attn_bwd_dq_bx3_kernel": "attn_bwd_dq_bx3_kernel",
             "attn_bwd_dq2_bx3_kernel": "attn_bwd_dq2_bx3_kernel", "attn_bwd1_bx3_kernel": "attn_bwd1_bx3_kernel",
             "fe_qkv_fwd_kernel": "fe_qkv_fwd_kernel",
             "fe_fwd_a_kernel": "fe_fwd_a_kernel", "fe_fwd_b_kernel": "fe_fwd_b_kernel", "fe_bwd_a_kernel": "fe_bwd_a_kernel",
             "fe_bwd_b_kernel": "fe_bwd_b_kernel", "fe_bwd_qkv_kernel": "fe_bwd_qkv_kernel"}
    out["step_total"] = {"fetch_kb": sum(t for _, t in fetch.values()), "write_kb": sum(t for _, t in write.values()),
                         "note": "all kernels of the profiled run (13 steps: 5 warm-up + 8 timed): bytes = (2 * fetch_kb + write_kb) * 1024"}
    for key, sub in short.items():
        f = [(n, t) for k, (n, t) in fetch.items() if sub in k]
        w = [(n, t) for k, (n, t) in write.items() if sub in k]
        if not f or not w:
            continue
        nf, tf = sum(a for a, _ in f), sum(b for _, b in f)
        nw, tw = sum(a for a, _ in w), sum(b for _, b in w)
        out["kernels"][key] = {"launches": nf, "fetch_kb_per_launch": tf / nf, "write_kb_per_launch": tw / nw,
                               "bytes_per_launch": (2.0 * tf / nf + tw / nw) * 1024.0}
    dst = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "profiles", "r02_pmc_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["kernels"].get("conv3x3_halo_kernel")))


if __name__ == "__main__":
    main()
