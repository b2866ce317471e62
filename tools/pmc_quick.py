"""Per-kernel averages of one rocprofv3 --pmc pass (counter_collection.csv): duration, clock, MFMA-pipe busy fraction,
wait fractions.   python tools/pmc_quick.py <counter_collection.csv> [substring ...]"""
import csv
import re
import sys
from collections import defaultdict

path, only = sys.argv[1], sys.argv[2:]
RAW = "--raw" in only
only = [o for o in only if o != "--raw"]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
seen = set()
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    if only and not any(o in name for o in only):
        continue
    key = re.sub(r"\(.*", "", name)[:90] + f" grid{r.get('Grid_Size', '')}"
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    c = {n: sum(v) / len(v) for n, v in cs.items()}
    us = sum(dur[k]) / len(dur[k])
    line = f"{k}\n   n={len(dur[k])} {us:8.1f} us"
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    if cyc:
        line += f"  clock {cyc / (us * 1e3):.2f} GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            line += f"  mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f}"
    if c.get("SQ_WAVE_CYCLES"):
        w = c["SQ_WAVE_CYCLES"]
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if n in c:
                line += f"  {n[3:]}/wave {c[n] / w:.3f}"
    if c.get("SQ_INSTS_MFMA"):
        line += f"  valu/mfma {(c.get('SQ_INSTS_VALU', 0) - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA']:.2f}"
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        line += f"  lds_conflict {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}"
    for n in ("FETCH_SIZE", "WRITE_SIZE"):
        if n in c:
            line += f"  {n} {c[n] / 1024:.1f} MiB"
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_WAVE_CYCLES"):
        line += f"  WAVE_CYCLES/BUSY_CYCLES {c['SQ_WAVE_CYCLES'] / c['SQ_BUSY_CYCLES']:.2f}"
    if RAW:
        line += "\n   " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items()))
    print(line)
