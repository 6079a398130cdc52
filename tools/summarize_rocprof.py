"""Summarise rocprofv3 (rocpd sqlite) outputs under gpurun_out/ into small text files under profiles/.

usage: python tools/summarize_rocprof.py r02 [workload ...]      (default workloads: headline c2 c4 c5)
Writes profiles/<round>_rocprof_<workload>.txt and merges the HBM traffic of each workload's dominant kernel into
profiles/traffic.json (2*FETCH_SIZE + WRITE_SIZE; the x2 is the gfx950 FETCH_SIZE correction of
MI355X_MICROARCH.md, section HBM).
"""
import glob, json, os, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
workloads = sys.argv[2:] or ["headline", "c2", "c4", "c5"]
DOMINANT = {"headline": "sweep_dma_kernel", "c2": "sweep_dma_kernel", "c3": "sweep_dma_kernel", "c4": "joint_kernel",
            "c5": "traj_eval_kernel", "i8": "sweep_i8_kernel", "auto": "sweep_i8_kernel", "i8x5": "sweep_i8_kernel"}
# on the GPU box: SUMMARY_DIR=gpurun_out (the raw databases are too big to travel back); locally: profiles/
out_dir = os.environ.get("SUMMARY_DIR") or os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)
tfile = os.path.join(out_dir, "traffic.json")
if not os.path.exists(tfile) and os.path.exists(os.path.join(ROOT, "profiles", "traffic.json")):
    import shutil
    shutil.copy(os.path.join(ROOT, "profiles", "traffic.json"), tfile)
traffic_all = json.load(open(tfile)) if os.path.exists(tfile) else {}
for w in workloads:
    lines, traffic = [], {}
    for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_{w}_*", "*_results.db"))):
        tag = os.path.basename(os.path.dirname(path))
        cur = sqlite3.connect(path).cursor()
        try:
            rows = list(cur.execute("select * from top_kernels limit 12"))
        except sqlite3.Error as e:
            lines.append(f"==== {tag}: unreadable ({e})")
            continue
        lines.append(f"==== {tag}: rocprofv3 --kernel-trace (view top_kernels; durations in us)")
        lines.append(f"{'kernel':100s} {'calls':>6s} {'total_us':>14s} {'avg_us':>14s} {'pct':>7s}")
        for name, calls, total, avg, pct in rows:
            lines.append(f"{name[:100]:100s} {calls:6d} {total:14.0f} {avg:14.0f} {pct:7.2f}")
        try:
            rows = list(cur.execute(
                "select kernel_name, counter_name, sum(value), count(*), avg(duration), max(grid_size), max(workgroup_size), "
                "max(lds_block_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count) from counters_collection "
                f"where kernel_name like '%{DOMINANT.get(w, 'sweep')}%' group by kernel_name, counter_name, dispatch_id"))
        except sqlite3.Error:
            rows = []
        if rows:
            lines.append(f"-- PMC counters, {DOMINANT.get(w)} dispatches ({tag})")
            for r in rows:
                lines.append(f"   {r[0][:60]:60s} {r[1]:28s} sum={r[2]:.6g} (n={r[3]}) dur_ns={r[4]:.0f} grid={r[5]} wg={r[6]} "
                             f"lds={r[7]} vgpr={r[8]} agpr={r[9]} sgpr={r[10]}")
                if r[1] in ("FETCH_SIZE", "WRITE_SIZE"):
                    traffic[r[1]] = traffic.get(r[1], 0.0) + r[2] * 1024.0  # counters are in KiB; summed over launches of a step
        lines.append("")
    if not lines:
        continue
    open(os.path.join(out_dir, f"{rnd}_rocprof_{w}.txt"), "w").write("\n".join(lines))
    print("\n".join(lines))
    if traffic:
        # the profiled command runs the bench's untimed first call + 1 timed step = 2 steps: per-step traffic = half
        traffic = {k: v / 2.0 for k, v in traffic.items()}
        traffic_all[w] = traffic.get("FETCH_SIZE", 0.0) * 2.0 + traffic.get("WRITE_SIZE", 0.0)
        traffic_all.setdefault("_raw", {})
        if not isinstance(traffic_all["_raw"], dict) or "FETCH_SIZE" in traffic_all["_raw"]:
            traffic_all["_raw"] = {"headline_r01": traffic_all["_raw"]}
        traffic_all["_raw"][f"{w}_{rnd}"] = traffic
        traffic_all["_round"] = rnd
        traffic_all.setdefault("_rounds", {})[w] = rnd   # the round EVERY entry was measured in (bench.py prints it)
        traffic_all["_note"] = ("HBM bytes per step of the workload's dominant kernel = 2*FETCH_SIZE + WRITE_SIZE (KiB->bytes; the x2 "
                                "on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE uncalibrated); "
                                "the profiled run = the bench's untimed first call + 1 timed step, so the sums over dispatches are halved")
json.dump(traffic_all, open(tfile, "w"), indent=1)
