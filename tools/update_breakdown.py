"""Summarise gpurun_out/prof_upd (rocprofv3 --kernel-trace of tools/prof_update.py) into profiles/<round>_update_breakdown.txt."""
import glob, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
path = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_upd", "*_results.db"))[0]
cur = sqlite3.connect(path).cursor()
out = ["`update` + NLML/gradient at N=4096, d=8, matern52: rocprofv3 --kernel-trace of tools/prof_update.py (3 rounds).",
       "Durations of back-to-back dependent kernels include the ~5 us dispatch/flush floor of this part.", "",
       f"{'kernel':60s} {'tiles':>9s} {'calls/round':>11s} {'avg_us':>9s} {'us/round':>10s}"]
q = ("select substr(name,1,60), grid_x/workgroup_x, grid_y/workgroup_y, count(*)/3.0, avg(duration)/1e3, sum(duration)/3e3 "
     "from kernels group by 1,2,3 order by 6 desc")
tot = 0.0
for name, gx, gy, calls, avg, per in cur.execute(q):
    out.append(f"{name:60s} {gx:4d}x{gy:<4d} {calls:11.1f} {avg:9.1f} {per:10.1f}")
    tot += per
out.append(f"{'total':60s} {'':9s} {'':11s} {'':9s} {tot:10.1f}")
open(os.path.join(ROOT, "profiles", f"{rnd}_update_breakdown.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
