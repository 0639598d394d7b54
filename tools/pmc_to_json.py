#!/usr/bin/env python3
"""Turn the three rocprofv3 PMC passes of tools/pmc_round.sh into the stored-traffic record bench.py cites
(profiles/pmc_trunk.json).  usage: pmc_to_json.py OUT.json MFMA.db FETCH.db WRITE.db COMMIT

HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE (KB; the gfx950 correction of
MI355X_MICROARCH.md "HBM": wide coalesced streaming reads are tallied at half their bytes) + WRITE_SIZE (KB)."""
import json
import sqlite3
import sys

KERNEL = "trunk_infer_kernel"


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
         "where counter_name = ? group by kernel_name")
    for name, n, v, d in cur.execute(q, (counter,)):
        if KERNEL in name and "x3" not in name:
            return n, v, d
    return None


def main():
    out, mfma, fetch, write, commit = sys.argv[1:6]
    B = N = 1024
    rec = {"kernel": KERNEL, "config": "B=1024 N=1024 fp32 (BASELINE configs[1])", "B": B, "N": N, "commit": commit,
           "source": "tools/pmc_round.sh: rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE and the SQ_* set "
                     "each in its own run, --kernel-trace only)"}
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    if f and w:
        rec.update(fetch_size_kb_raw=round(f[1], 2), write_size_kb=round(w[1], 2),
                   fetch_correction="x2 (gfx950 FETCH_SIZE reports 1/2 of streamed read bytes, MI355X_MICROARCH.md HBM "
                                    "section); WRITE_SIZE taken as is",
                   traffic_bytes_per_launch=int(round((2 * f[1] + w[1]) * 1024)),
                   algorithmic_bytes_per_launch=B * N * 12 + B * 1024 * 4,
                   avg_launch_us_under_fetch_pass=round(f[2] / 1e3, 1))
    busy, act, conf = (per_kernel(mfma, c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT"))
    if busy and act:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs:
        # busy fraction = busy / (active cycles of one XCD x 1024)   (round-1 record: 4563.4e6 / (39.57e6/8 x 1024) = 0.901)
        rec.update(sq_valu_mfma_busy_cycles=busy[1], grbm_gui_active_all_xcd=act[1],
                   mfma_busy_frac=round(busy[1] / (act[1] / 8 * 1024), 3),
                   avg_launch_us_under_sq_pass=round(busy[2] / 1e3, 1))
    if conf:
        rec["sq_lds_bank_conflict"] = conf[1]
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
