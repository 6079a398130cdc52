import sqlite3, sys, glob
for path in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    cur = sqlite3.connect(path).cursor()
    for name, calls, total, avg, pct in cur.execute("select * from top_kernels limit 16"):
        print(f"{name[:90]:90s} {calls:6d} {total:12.0f} {avg:10.1f} {pct:6.2f}")
