import csv, glob, sys, collections
path = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
rows=[r for r in csv.DictReader(open(path)) if r["Counter_Name"]=="FETCH_SIZE"]
rows.sort(key=lambda r:int(r["Dispatch_Id"]))
for r in rows: print(r["Dispatch_Id"], r["Kernel_Name"][:16], r["Counter_Value"])
