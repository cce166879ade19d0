import csv,glob,sys
for d in sys.argv[1:]:
    f=glob.glob(d+"/*/*_kernel_trace.csv")[0]
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    t0=int(rows[0]["Start_Timestamp"])
    print(d)
    for r in rows:
        n=r["Kernel_Name"].split("(")[0][-28:]
        if any(k in n for k in ("horner","ml_small","fexp_team","lane2x","lines4","clear","k_pow","swj")):
            print("  %-30s start %10.3f ms dur %8.3f ms  wg %s grid %s" % (n,(int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r.get("Workgroup_Size_X"), r.get("Grid_Size_X")))
