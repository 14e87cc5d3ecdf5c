import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ''
f = glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:60]
    if pat and pat not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in acc.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f'    {name:32s} {v:.4g}')
