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
# idle time between consecutive kernels of the LAST `steps`-th of the trace (one steady-state step): where the queue ran dry
evs = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'^void ', '', re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])).split('(')[0][:40]) for r in rows))
evs = evs[-(len(evs) // steps):]
gaps, busy_end = [], evs[0][1]
for (s, e, n), (ps, pe, pn) in zip(evs[1:], evs[:-1]):
    if s > busy_end:
        gaps.append(((s - busy_end) / 1e3, pn, n))
    busy_end = max(busy_end, e)
span = (evs[-1][1] - evs[0][0]) / 1e6
print(f'# last step: span {span:.2f} ms, idle between kernels {sum(g[0] for g in gaps)/1e3:.2f} ms in {len(gaps)} gaps; largest:')
for g, pn, n in sorted(gaps, reverse=True)[:12]:
    print(f'#   {g:8.1f} us  after {pn:40s} before {n}')
by = collections.defaultdict(float)
for g, pn, n in gaps:
    by[n] += g
print('# idle by following kernel: ' + ', '.join(f'{k} {v/1e3:.2f} ms' for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]))
