"""Summarise rocprofv3 (rocpd sqlite) outputs under gpurun_out/ into small text files under profiles/.

usage: python tools/summarize_rocprof.py r01
"""
import glob, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)
lines = []
traffic = {}
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_*", "*_results.db"))):
    tag = os.path.basename(os.path.dirname(path))
    db = sqlite3.connect(path)
    cur = db.cursor()
    lines.append(f"==== {tag}: rocprofv3 --kernel-trace --stats equivalent (view top_kernels; durations in us)")
    lines.append(f"{'kernel':100s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'pct':>7s}")
    for name, calls, total, avg, pct in cur.execute("select * from top_kernels limit 14"):
        lines.append(f"{name[:100]:100s} {calls:6d} {total:14.0f} {avg:14.0f} {pct:7.2f}")
    rows = list(cur.execute(
        "select kernel_name, counter_name, sum(value), count(*), avg(duration), max(grid_size), max(workgroup_size), "
        "max(lds_block_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count) from counters_collection "
        "where kernel_name like '%sweep%' group by kernel_name, counter_name, dispatch_id"))
    if rows:
        lines.append(f"-- PMC counters, sweep kernel dispatches ({tag})")
        for r in rows:
            lines.append(f"   {r[0][:60]:60s} {r[1]:28s} sum={r[2]:.6g} (n={r[3]}) dur_ns={r[4]:.0f} grid={r[5]} wg={r[6]} "
                         f"lds={r[7]} vgpr={r[8]} agpr={r[9]} sgpr={r[10]}")
            if r[1] in ("FETCH_SIZE", "WRITE_SIZE"):
                traffic[r[1]] = r[2] * 1024.0  # counters are in KiB
    lines.append("")
txt = "\n".join(lines)
open(os.path.join(out_dir, f"{rnd}_rocprof_summary.txt"), "w").write(txt)
print(txt)
if traffic:
    # MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x
    fetch = traffic.get("FETCH_SIZE", 0.0) * 2.0
    write = traffic.get("WRITE_SIZE", 0.0)
    info = {"headline": fetch + write,
            "_note": "HBM bytes per sweep launch = 2*FETCH_SIZE + WRITE_SIZE (KiB->bytes; the x2 on FETCH_SIZE is the "
                     "gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE uncalibrated)",
            "_raw": traffic, "_round": rnd}
    json.dump(info, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)
    print(json.dumps(info))
