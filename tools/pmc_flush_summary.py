import csv, sys, glob, collections
for tag, path in (("FETCH_SIZE", sys.argv[1]), ("WRITE_SIZE", sys.argv[2])):
    files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == tag:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        if "flush" in k or "gain_delayed" in k or "k_predict" in k:
            mult = 2 if tag == "FETCH_SIZE" else 1
            print(f"{tag} {k:48s} launches {len(v):4d} avg {sum(v)/len(v):16.1f} KiB -> {sum(v)/len(v)*1024*mult/1e9:8.2f} GB per launch" + (" (x2: gfx950 FETCH_SIZE correction)" if mult == 2 else ""))
