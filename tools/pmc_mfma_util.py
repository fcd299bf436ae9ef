#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from a tools/rocpd_pmc.py table of a
`rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.
rocprofv3 stores one GRBM_GUI_ACTIVE record per XCD (8 per dispatch) and one SQ_VALU_MFMA_BUSY_CYCLES record per
XCD x shader-engine (32 per dispatch):
    util = (sum of busy cycles / dispatches) / (mean GUI_ACTIVE per record x 1024 SIMDs),  dispatches = GUI records / 8.
usage: python tools/pmc_mfma_util.py profiles/rXX_pmc_mfma.csv [min_util]"""
import csv
import sys


def main():
    busy, gui = {}, {}
    for r in csv.DictReader(open(sys.argv[1])):
        d = busy if r["counter"] == "SQ_VALU_MFMA_BUSY_CYCLES" else gui if r["counter"] == "GRBM_GUI_ACTIVE" else None
        if d is not None:
            d[r["kernel"]] = (float(r["sum"]), int(r["dispatches"]))
    floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
    rows = []
    for k, (b, _) in busy.items():
        if k not in gui or b <= 0:
            continue
        gsum, grec = gui[k]
        disp = grec / 8.0
        util = (b / disp) / ((gsum / grec) * 1024.0)
        rows.append((util, k, disp, b / disp, gsum / grec))
    print("%-78s %9s %12s %12s %8s" % ("kernel", "launches", "busy/launch", "GUI_ACTIVE", "util"))
    for util, k, disp, bpl, g in sorted(rows, reverse=True):
        if util >= floor:
            print("%-78s %9d %12.4g %12.4g %8.3f" % (k[:78], disp, bpl, g, util))


if __name__ == "__main__":
    main()
