"""Print the tail of a rocprofv3 kernel + memory-copy trace as one timeline (design aid)."""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ev = []
for f in glob.glob(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                   r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '')[:40]))
for f in glob.glob(d + "/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
ev.sort()
ev = [e for e in ev if 'at::native' not in e[2] and 'rocclr' not in e[2]]
t0 = ev[-n][0]
prev = None
for s, e, name in ev[-n:]:
    print(f"{(s-t0)/1e3:9.1f} -> {(e-t0)/1e3:9.1f} us ({(e-s)/1e3:7.1f})  {name}")
