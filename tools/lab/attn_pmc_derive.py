"""Derived pipe-utilisation figures per (kernel, grid) from the rocprofv3 --pmc passes of tools/lab/attn_pmc.sh.
Units (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE is summed over the 8 XCDs (active shader cycles = / 8); SQ_VALU_MFMA_BUSY_CYCLES counts cycles
summed over the 1024 SIMDs; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count QUAD-cycles summed over all waves.
    python tools/lab/attn_pmc_derive.py /tmp/ap1 /tmp/ap2"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"][:60], r.get("Grid_Size", ""))
            acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[key][r["Counter_Name"]] += 1
SIMDS = 1024.0
for key in sorted(acc):
    c = {k: acc[key][k] / cnt[key][k] for k in acc[key]}
    if "GRBM_GUI_ACTIVE" not in c or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    act = c["GRBM_GUI_ACTIVE"] / 8.0
    simd_cycles = SIMDS * act
    q = lambda n: 4.0 * c.get(n, 0.0)                 # quad-cycles -> cycles
    print(f"{key[0]} grid {key[1]}: {act:,.0f} active shader cycles per dispatch")
    print(f"    matrix pipe busy              {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:5.1f} % of SIMD-cycles")
    print(f"    VALU issuing (ACTIVE_INST_VALU){100 * q('SQ_ACTIVE_INST_VALU') / simd_cycles:5.1f} % of SIMD-cycles  "
          f"({c.get('SQ_ACTIVE_INST_VALU', 0) / max(c.get('SQ_INSTS_VALU', 1), 1):.2f} quad-cycles per VALU instruction)")
    print(f"    matrix + VALU                 {100 * (c['SQ_VALU_MFMA_BUSY_CYCLES'] + q('SQ_ACTIVE_INST_VALU')) / simd_cycles:5.1f} %  (100 % = no cycle in which a SIMD runs neither)")
    print(f"    LDS instructions issuing      {100 * q('SQ_ACTIVE_INST_LDS') / simd_cycles:5.1f} %, VMEM {100 * q('SQ_ACTIVE_INST_VMEM') / simd_cycles:5.1f} %, "
          f"VMEM instruction cycles {100 * q('SQ_INST_CYCLES_VMEM') / simd_cycles:5.1f} %")
    wc = q("SQ_WAVE_CYCLES")
    if wc:
        print(f"    resident waves per SIMD       {wc / simd_cycles:5.2f}; of the wave-cycles: waiting for an instruction's operands / issue "
              f"(WAIT_INST_ANY) {100 * q('SQ_WAIT_INST_ANY') / wc:4.1f} %, waiting on s_waitcnt (WAIT_ANY) {100 * q('SQ_WAIT_ANY') / wc:4.1f} %, "
              f"issuing (ACTIVE_INST_ANY) {100 * q('SQ_ACTIVE_INST_ANY') / wc:4.1f} %")
    print(f"    LDS bank conflict cycles      {c.get('SQ_LDS_BANK_CONFLICT', 0):,.0f}")
