#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite output into the small summaries committed under profiles/.

  python tools/rocpd_summary.py stats  <results.db> <out.csv>            # per-kernel calls/total/avg/min/max (ns)
  python tools/rocpd_summary.py pmc    <results.db> <kernel-substring>   # per-launch counter values of one kernel
  python tools/rocpd_summary.py traffic <fetch.db> <write.db> <kernel-substring> <M> <N> <K> <out.json>
  python tools/rocpd_summary.py pmcjson <out.json> <kernel-substring> <pass1.db> [<pass2.db> ...]
        # one JSON per kernel: mean of every counter over its launches (first launch of each pass dropped), mean
        # dispatch duration per pass, register counts, and the utilisation figures derived from them
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


def pmcjson(out, sub, dbs):
    from aaclip_hip._lib import kernel_source_revision
    doc = {"kernel_substring": sub, "counters": {}, "passes": [],
           "kernel_revision": kernel_source_revision(("attention.hip", "gemm256t.hip", "gemm.hip", "common.h",
                                                      "mma16.h", "kernels.h"))}
    # "<substring>@i/n": of the matching dispatches (in launch order) only every n-th, starting at the i-th -- two
    # products that share one kernel instantiation (out_proj and c_proj: both EPI_BIAS_RESID, launched alternately)
    pick = None
    if "@" in sub:
        sub, sel = sub.rsplit("@", 1)
        pick = tuple(int(v) for v in sel.split("/"))
        doc["kernel_substring"] = sub
        doc["dispatch_selection"] = f"dispatch {pick[0]} of every {pick[1]} matching launches, in launch order"
    for path in dbs:
        db = sqlite3.connect(path)
        rows = db.execute("select dispatch_id, counter_name, value, duration, kernel_name, vgpr_count, "
                          "accum_vgpr_count, sgpr_count, lds_block_size, grid_size, workgroup_size "
                          "from counters_collection where kernel_name like ? order by dispatch_id",
                          (f"%{sub}%",)).fetchall()
        if pick and rows:
            order = {d: i for i, d in enumerate(sorted({r[0] for r in rows}))}
            rows = [r for r in rows if order[r[0]] % pick[1] == pick[0]]
        if not rows:
            doc["passes"].append({"db": os.path.basename(path), "launches": 0})
            continue
        first = rows[0][0]
        keep = [r for r in rows if r[0] != first] or rows
        per = {}
        for r in keep:
            per.setdefault(r[1], []).append(r[2])
        durs = {}
        for r in keep:
            durs[r[0]] = r[3]
        doc["kernel_name"] = keep[0][4][:200]
        doc["registers"] = {"vgpr": keep[0][5], "accum_vgpr": keep[0][6], "sgpr": keep[0][7], "lds_bytes": keep[0][8],
                            "grid": keep[0][9], "workgroup": keep[0][10]}
        doc["passes"].append({"db": os.path.basename(path), "launches": len(durs), "counters": sorted(per),
                              "mean_duration_us_while_profiled": round(sum(durs.values()) / len(durs) / 1e3, 2)})
        for k, v in per.items():
            doc["counters"][k] = sum(v) / len(v)
    c = doc["counters"]
    d = {}
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0          # rocprofv3 sums the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
        d["gpu_active_cycles"] = cyc
        simd = cyc * 256 * 4
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd
        if "SQ_ACTIVE_INST_VALU" in c:
            d["valu_issue_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / simd     # quad-cycles
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
            d["valu_mfma_coexec_frac"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / simd
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS"):
            if k in c:
                d[k.lower() + "_per_wave_cycle"] = c[k] / c["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac_of_lds_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c:
        d["hbm_read_bytes_per_launch_fetch_x2"] = c["FETCH_SIZE"] * 1024 * 2   # gfx950 correction (guide, HBM section)
    if "WRITE_SIZE" in c:
        d["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    doc["derived"] = d
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    print(json.dumps(d, indent=1))


def main():
    mode = sys.argv[1]
    if mode == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif mode == "pmc":
        for name, vals in counter_values(sys.argv[2], sys.argv[3]).items():
            print(name, len(vals), "launches, mean", sum(vals) / len(vals))
    elif mode == "pmcjson":
        pmcjson(sys.argv[2], sys.argv[3], sys.argv[4:])
    elif mode == "traffic":
        fetch_db, write_db, sub, M, N, K, out = sys.argv[2:9]
        precision = sys.argv[9] if len(sys.argv) > 9 else "fp16"
        clip_weights = sys.argv[10] if len(sys.argv) > 10 else "fp32"
        M, N, K = int(M), int(N), int(K)
        # operand bytes of the arithmetic mode: fp16 = 2 per element; fp16x2 = split rows, 4 per element (3 for the A
        # and W planes that are actually read when the weight is exact in fp16: the hi8 / Wl8 planes are not touched)
        if precision == "fp16x2":
            ab = 3 if clip_weights == "fp16" else 4
            algo = M * K * ab + N * K * ab + M * N * 4
        else:
            algo = M * K * 2 + N * K * 2 + M * N * 2
        f = counter_values(fetch_db, sub)["FETCH_SIZE"]
        w = counter_values(write_db, sub)["WRITE_SIZE"]
        f_kb, w_kb = sum(f) / len(f), sum(w) / len(w)
        from aaclip_hip._lib import kernel_source_revision
        doc = {
            "kernel": sub, "shape": [M, N, K], "launches": [len(f), len(w)], "precision": precision,
            "clip_weights": clip_weights,
            "kernel_revision": kernel_source_revision(),   # bench.py refuses this file once the kernel sources change
            "FETCH_SIZE_raw_kb": f_kb, "WRITE_SIZE_raw_kb": w_kb,
            "fetch_bytes_corrected_x2": f_kb * 1024 * 2, "write_bytes": w_kb * 1024,
            "traffic_bytes_per_launch": f_kb * 1024 * 2 + w_kb * 1024,
            "algorithmic_bytes_per_launch": algo,
            "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)",
        }
        with open(out, "w") as fh:
            json.dump(doc, fh, indent=1)
        print(json.dumps(doc))


if __name__ == "__main__":
    main()
