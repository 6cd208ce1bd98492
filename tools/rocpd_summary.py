#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite output into the small summaries committed under profiles/.

  python tools/rocpd_summary.py stats  <results.db> <out.csv>            # per-kernel calls/total/avg/min/max (ns)
  python tools/rocpd_summary.py pmc    <results.db> <kernel-substring>   # per-launch counter values of one kernel
  python tools/rocpd_summary.py traffic <fetch.db> <write.db> <kernel-substring> <M> <N> <K> <out.json>
"""
import csv
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "aa-clip-iqm_amd"))


def stats(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, n, tot, avg, mn, mx in rows:
            w.writerow([name[:160], n, int(tot), round(avg, 1), round(100.0 * tot / total, 3), int(mn), int(mx)])
    print(f"{len(rows)} kernels, {total / 1e6:.2f} ms of kernel time -> {out_csv}")


def counter_values(db_path, kernel_sub):
    db = sqlite3.connect(db_path)
    rows = db.execute("select counter_name, value from counters_collection where kernel_name like ?",
                      (f"%{kernel_sub}%",)).fetchall()
    out = {}
    for name, v in rows:
        out.setdefault(name, []).append(v)
    return out


def main():
    mode = sys.argv[1]
    if mode == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif mode == "pmc":
        for name, vals in counter_values(sys.argv[2], sys.argv[3]).items():
            print(name, len(vals), "launches, mean", sum(vals) / len(vals))
    elif mode == "traffic":
        fetch_db, write_db, sub, M, N, K, out = sys.argv[2:9]
        M, N, K = int(M), int(N), int(K)
        f = counter_values(fetch_db, sub)["FETCH_SIZE"]
        w = counter_values(write_db, sub)["WRITE_SIZE"]
        f_kb, w_kb = sum(f) / len(f), sum(w) / len(w)
        from aaclip_hip._lib import kernel_source_revision
        doc = {
            "kernel": sub, "shape": [M, N, K], "launches": [len(f), len(w)],
            "kernel_revision": kernel_source_revision(),   # bench.py refuses this file once the kernel sources change
            "FETCH_SIZE_raw_kb": f_kb, "WRITE_SIZE_raw_kb": w_kb,
            "fetch_bytes_corrected_x2": f_kb * 1024 * 2, "write_bytes": w_kb * 1024,
            "traffic_bytes_per_launch": f_kb * 1024 * 2 + w_kb * 1024,
            "algorithmic_bytes_per_launch": M * K * 2 + N * K * 2 + M * N * 2,
            "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)",
        }
        with open(out, "w") as fh:
            json.dump(doc, fh, indent=1)
        print(json.dumps(doc))


if __name__ == "__main__":
    main()
