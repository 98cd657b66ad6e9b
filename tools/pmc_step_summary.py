"""Whole-step HBM traffic from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs).
usage: pmc_step_summary.py DIR STEPS   (DIR/FETCH_SIZE/*counter_collection.csv, DIR/WRITE_SIZE/*counter_collection.csv)
Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KiB-like units of 1 KB = 1024 B by
rocprofv3's derived metric; on gfx950 FETCH_SIZE reports HALF of the bytes of wide (16 B/lane) coalesced streaming reads, so
it is doubled here; WRITE_SIZE is taken as is (uncalibrated)."""
import collections, csv, glob, json, re, sys
d, steps = sys.argv[1], int(sys.argv[2])
out = {}
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    tot = 0.0
    for f in glob.glob(f'{d}/{c}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] != c:
                continue
            v = float(row['Counter_Value'])
            tot += v
            name = re.sub(r'\(anonymous namespace\)::', '', row['Kernel_Name']); name = re.sub(r'^void ', '', name).split('(')[0][:60]
            per_kernel[name][c] += v
    out[c] = tot / steps
fetch_b = out['FETCH_SIZE'] * 1024 * 2      # x2: gfx950 correction for wide streaming reads
write_b = out['WRITE_SIZE'] * 1024
print(f'# whole-step HBM traffic (avg over {steps} steps incl. warm-up): FETCH_SIZE {out["FETCH_SIZE"]:.0f} KB x2 -> {fetch_b/1e9:.2f} GB read, '
      f'WRITE_SIZE {out["WRITE_SIZE"]:.0f} KB -> {write_b/1e9:.2f} GB written, total {(fetch_b+write_b)/1e9:.2f} GB/step')
print(json.dumps({'fetch_bytes_per_step': fetch_b, 'write_bytes_per_step': write_b, 'hbm_bytes_per_step': fetch_b + write_b}))
print(f'{"kernel":62s} {"read GB/step":>13s} {"write GB/step":>14s}')
for k, v in sorted(per_kernel.items(), key=lambda kv: -(kv[1]['FETCH_SIZE'] * 2 + kv[1]['WRITE_SIZE']))[:25]:
    print(f'{k:62s} {v["FETCH_SIZE"]*2048/steps/1e9:13.3f} {v["WRITE_SIZE"]*1024/steps/1e9:14.3f}')
