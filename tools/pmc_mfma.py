#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 SQ counter pass over tools/bin/conv_bench:
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
            --kernel-trace --output-format csv -d out -o m -- tools/bin/conv_bench 3 64
  python tools/pmc_mfma.py out/m_counter_collection.csv profiles/r01_conv_mfma_util.csv
util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the busy counter is summed over all SIMDs and counts
cycles (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md 'cycle constants'); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv
import sys
from collections import Counter, defaultdict

src, dst = sys.argv[1:3]
shape = sys.argv[3] if len(sys.argv) > 3 else ""          # label of the conv_bench shape this pass ran (CB_ONLY)
agg, n, grid = defaultdict(lambda: defaultdict(float)), Counter(), {}
for r in csv.DictReader(open(src)):
    k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"], r["LDS_Block_Size"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[k] += 1
with open(dst, "w") as f:
    f.write("shape,kernel,grid_size,lds_bytes,launches,mfma_busy_cycles_per_launch,active_cycles_per_launch,mfma_util,wait_any_frac,wait_inst_any_frac,active_inst_frac\n")
    for k, v in agg.items():
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
            continue
        act = v["GRBM_GUI_ACTIVE"] / 8.0
        wc = max(v["SQ_WAVE_CYCLES"], 1.0)
        row = (k[0], k[1], k[2], n[k], v["SQ_VALU_MFMA_BUSY_CYCLES"] / n[k], act / n[k], v["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * 1024),
               v["SQ_WAIT_ANY"] / wc, v["SQ_WAIT_INST_ANY"] / wc, v["SQ_ACTIVE_INST_ANY"] / wc)
        f.write("%s,\"%s\",%s,%s,%d,%.4e,%.4e,%.4f,%.3f,%.3f,%.3f\n" % ((shape,) + row))
        print("%-50s grid %9s n=%3d util %.3f  wait_any %.2f wait_inst %.2f active %.2f" % (k[0][-50:], k[1], n[k], row[6], row[7], row[8], row[9]))
