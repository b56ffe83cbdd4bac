"""Dev tool: per-kernel means of the SQ counter passes under gpurun_out/sq_<tag>_*/ (tools/run_sq*.sh)."""
import collections, csv, glob, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/sq_%s_*/sq_counter_collection.csv' % tag)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'inflate' not in k and (len(sys.argv) < 3 or sys.argv[2] not in k):
            continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, d in acc.items():
    print(k, 'mean dispatch %.3f ms' % (sum(dur[k]) / len(dur[k])))
    for c in sorted(d):
        v = d[c]
        print('   %-24s %.4g' % (c, sum(v) / len(v)))
