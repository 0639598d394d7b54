#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into a small text file for profiles/.

usage: tools/rocprof_summary.py OUT.md LABEL=path/to/results.db [LABEL=...]
Kernel-trace dbs yield the --stats style table (calls, total, average); PMC dbs yield the
per-kernel average of every collected counter.  Only kernels of this repo (pngpd) are listed
in full; everything else is folded into one 'other' line.
"""
import sqlite3
import sys

OURS = ("trunk_", "fc_", "pool_", "fold_conv_bn", "crop_", "resample", "bn", "pngpd", "gpg_", "hand_box", "split_pack",
        "cloud_moments", "a_cvec", "bwd_e_prep", "dtrans", "dw1_", "dw3_", "log_softmax_bwd", "reduce_partials", "adam_",
        "batch_keep_rows", "block_scan", "conv5_", "depth_", "gpd_", "mfma_rate", "nll_", "reduce_fin", "relu_bwd",
        "stack_gather_lists", "train_pack")


def short(name):
    n = name.split("(")[0]
    return n[5:] if n.startswith("void ") else n


def main():
    args = sys.argv[1:]
    show_all = "--all" in args          # list foreign (torch / rocclr) kernels individually too
    args = [a for a in args if a != "--all"]
    out, items = args[0], args[1:]
    lines = []
    for it in items:
        label, path = it.split("=", 1)
        cur = sqlite3.connect(path).cursor()
        lines.append(f"## {label}  ({path.split('gpurun_out/')[-1]})\n")
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        if rows:
            lines.append("| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|")
            other = [0, 0.0, 0.0]
            for n, c, t, a, p in rows:
                if show_all or short(n).startswith(OURS):
                    lines.append(f"| {short(n)} | {c} | {t:.1f} | {a:.3f} | {p:.2f} |")
                else:
                    other[0] += c; other[1] += t; other[2] += p
            lines.append(f"| (other: torch/rocclr) | {other[0]} | {other[1]:.1f} | - | {other[2]:.2f} |\n")
        try:
            q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                 "group by kernel_name, counter_name")
            rows = list(cur.execute(q))
        except sqlite3.Error:
            rows = []
        rows = [r for r in rows if short(r[0]).startswith(OURS)]
        if rows:
            lines.append("| kernel | counter | dispatches | avg value per dispatch | avg dispatch ns |\n|---|---|---|---|---|")
            for n, c, k, v, d in rows:
                lines.append(f"| {short(n)} | {c} | {k} | {v:.6g} | {d:.0f} |")
            lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
