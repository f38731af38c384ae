import sqlite3, glob, sys
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0])
pat = sys.argv[2] if len(sys.argv) > 2 else None
if pat:
    for r in db.execute("select grid_x, grid_y, count(*), avg(duration), min(duration) from kernels where name like ? group by grid_x, grid_y", ("%" + pat + "%",)): print(r)
else:
    for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 10"): print(r)
