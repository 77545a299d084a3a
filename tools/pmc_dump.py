import csv, collections, glob, sys
pat, key = sys.argv[1], sys.argv[2]
f = glob.glob(pat)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if key in r["Kernel_Name"]:
        d = agg.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"][:48], "grid": r["Grid_Size"], "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"]})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in agg.items():
    print(k, v)
