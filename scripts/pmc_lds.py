"""Scratch: sum PMC counters per kernel from a rocprofv3 counter_collection CSV."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row['Kernel_Name'][:60]
    acc[k][row['Counter_Name']] += float(row['Counter_Value'])
for k, d in acc.items():
    if 'k_tab_wpi' not in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s %.4g' % (c, v))
