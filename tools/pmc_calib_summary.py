import sqlite3, sys
for db, counter in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    kd, ks, pe, pi = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    q = f"""select s.kernel_name, d.dispatch_id, sum(e.value), count(*) from {pe} e join {kd} d on e.event_id = d.event_id
            join {ks} s on d.kernel_id = s.id join {pi} p on e.pmc_id = p.id where p.name = ? group by 1, 2 order by 2"""
    for name, disp, val, cnt in c.execute(q, (counter,)):
        print(counter, name.split("(")[0][:40], disp, "%.1f KiB" % val, "rows", cnt)
