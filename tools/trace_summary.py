"""Per-kernel / per-grid summary of a rocprofv3 --kernel-trace CSV. usage: trace_summary.py file.csv steps"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
agg = collections.defaultdict(list)
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); name = re.sub(r'^void ', '', name)
    short = name.split('(')[0][:66]
    key = (short, int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f'# total kernel time per step: {tot/steps/1e3:.2f} ms over {len(rows)/steps:.0f} dispatches/step')
print(f'{"kernel":68s} {"blocks":>7s} {"n/step":>7s} {"avg_us":>9s} {"ms/step":>8s} {"pct":>6s}')
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f'{k:68s} {g:7d} {len(v)/steps:7.1f} {sum(v)/len(v):9.1f} {sum(v)/steps/1e3:8.3f} {100*sum(v)/tot:6.2f}')
