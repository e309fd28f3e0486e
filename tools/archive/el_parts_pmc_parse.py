import csv, glob, os, sys, collections
root = sys.argv[1]
for kind, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = sorted(glob.glob(f"{root}/{kind}/*/*_counter_collection.csv"), key=os.path.getmtime)
    if not files: continue
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(files[-1])):
        if row["Counter_Name"] == name and "k_el_fast" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"][:40], row["Grid_Size"])].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        v = sorted(v); m = v[len(v)//2]
        print(kind, k, len(v), "median MB:", round((2 if kind == "fetch" else 1) * m * 1024 / 1e6, 1))
