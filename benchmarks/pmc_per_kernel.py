import csv, glob, sys, collections
csv.field_size_limit(1 << 30)
def tot(d, counter):
    import os
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == counter and "attn" in row["Kernel_Name"]:
            k = row["Kernel_Name"].split("(")[0][-48:]
            agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    return agg
base = sys.argv[1]
fe, wr = tot(base + "/bwd_FETCH_SIZE", "FETCH_SIZE"), tot(base + "/bwd_WRITE_SIZE", "WRITE_SIZE")
for k in fe:
    n = fe[k][0]
    print(k, n, "dispatches:", round((2 * fe[k][1] + wr[k][1]) * 1024 / n / 1e6, 2), "MB each (fetch x2 + write)")
