"""Per-(kernel, grid) averages of rocprofv3 PMC passes over the bench step.  usage: pmc_sq_summary.py DIR STEPS
DIR/p*/**/*counter_collection.csv.  SQ_* counters are summed over the chip's SEs by rocprofv3; GRBM_GUI_ACTIVE is summed over
the 8 XCDs (divide by 8 for cycles per dispatch); SQ_VALU_MFMA_BUSY_CYCLES counts per SIMD (x4 per CU), so
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)."""
import collections, csv, glob, re, sys
csv.field_size_limit(1 << 30)
d, steps = sys.argv[1], int(sys.argv[2])
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob(f'{d}/p*/**/*counter_collection.csv', recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        name = re.sub(r'\(anonymous namespace\)::', '', row['Kernel_Name']); name = re.sub(r'^void ', '', name).split('(')[0][:56]
        key = (name, int(row['Grid_Size']) // max(int(row['Workgroup_Size']), 1))
        tot[key][row['Counter_Name']] += float(row['Counter_Value']); cnt[(key, row['Counter_Name'])] += 1
        if row['Dispatch_Id'] not in seen:
            seen.add(row['Dispatch_Id']); dur[key].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
keys = sorted(tot, key=lambda k: -sum(dur[k]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]
for k in keys:
    c = {n: v / cnt[(k, n)] for n, v in tot[k].items()}
    us = sum(dur[k]) / len(dur[k])
    print(f'== {k[0]}  blocks {k[1]}  launches/step {len(dur[k]) / steps / max(1, len(glob.glob(d + "/p*/"))):.1f}  avg {us:.1f} us (serialised, counters on)')
    cyc = c.get('GRBM_GUI_ACTIVE', 0) / 8
    if cyc:
        print(f'   cycles/dispatch {cyc:.0f}  -> effective clock {cyc / us / 1e3:.2f} GHz')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in c: print(f'   MFMA busy  = {c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024):.3f} of SIMD-cycles')
        if 'SQ_ACTIVE_INST_VALU' in c: print(f'   VALU busy  = {c["SQ_ACTIVE_INST_VALU"] / (cyc * 1024):.3f} of SIMD-cycles (SQ_ACTIVE_INST_VALU / (cycles x 1024); MFMA issue included)')
        if 'SQ_LDS_IDX_ACTIVE' in c: print(f'   LDS busy   = {c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256):.3f} of CU-cycles; bank-conflict cycles {c.get("SQ_LDS_BANK_CONFLICT", 0) / (cyc * 256):.4f}')
    if c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0) > 0:
        print(f'   L2 hit rate {c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]):.3f}')
    for n, v in sorted(c.items()):
        print(f'      {n:28s} {v:18.1f}')
