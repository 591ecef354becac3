"""Per-kernel summary of a rocprofv3 results .db (kernel-trace): name, launches, avg / min / max us, grid, workgroup."""
import sqlite3
import sys

for db in sys.argv[1:]:
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), d.grid_size_x, "
         f"d.workgroup_size_x from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x "
         f"order by s.kernel_name, d.grid_size_x")
    print(db)
    for r in con.execute(q):
        n = r[0]
        if "ptgnn" in n:
            n = n.split("ptgnn_amd")[1]
        n = n.replace("12_GLOBAL__N_1", "")[:64]
        print(f"  {n:64s} n={r[1]:5d} avg={r[2] / 1e3:9.1f}us min={r[3] / 1e3:9.1f} max={r[4] / 1e3:9.1f} wgs={r[5] // max(r[6], 1)} wg={r[6]}")
