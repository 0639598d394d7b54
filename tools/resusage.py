#!/usr/bin/env python3
"""Condense `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stdin) into one line per kernel:
name, VGPRs(+AGPRs), SGPRs, spills, scratch, occupancy, LDS.  Used by `make resource-usage`."""
import re
import subprocess
import sys

rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+?)(?: \[[^\]]*\])?:\s+(\S+)", line)
    if not m:
        continue
    key, val = m.group(1).strip(), m.group(2)
    if key == "Function Name":
        cur = {"name": val}
        rows.append(cur)
    elif cur is not None:
        cur[key] = val
if rows:
    try:
        names = subprocess.run(["c++filt"] + [r["name"] for r in rows],
                               capture_output=True, text=True).stdout.split("\n")
    except Exception:
        names = [r["name"] for r in rows]
    for r, n in zip(rows, names):
        short = re.sub(r"\(.*", "", n.replace("void ", ""))
        print(f"{short:34s} vgpr {r.get('VGPRs', '?'):>3}+{r.get('AGPRs', '0'):<3} sgpr {r.get('TotalSGPRs', '?'):>3} "
              f"spill v{r.get('VGPRs Spill', '?')}/s{r.get('SGPRs Spill', '?')} scratch {r.get('ScratchSize', '?'):>4} "
              f"occ {r.get('Occupancy', '?')} lds {r.get('LDS Size', '?')}")
