"""MFMA-pipe utilisation per kernel from one rocprofv3 --pmc pass (CSV): counters SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over SIMDs),
   SQ_BUSY_CU_CYCLES (cycles, summed over CUs) + the kernel trace (durations).  pipe_busy = (MFMA_BUSY / (CUs*4)) / (BUSY_CU / CUs).
   The shader clock is NOT derived from these counters any more (BUSY_CU / duration gave 1.2-2.1 GHz: the quotient counts idle CUs and launch
   ramps as a slow clock; in-kernel s_memtime / s_memrealtime stamps measured 2.34-2.39 GHz inside the main loops of all three chip-filling
   kernels, profiles/r05_inkernel_clock.txt): frac_of_peak = pipe_busy x STAMPED_CLOCK / 2.4 GHz = the share of the 157.3 TFLOP/s peak the
   kernel's busy matrix-pipe cycles amount to."""
STAMPED_CLOCK_GHZ = 2.38
import collections, csv, re, sys
cc, kt = sys.argv[1], sys.argv[2]
CUS = 256
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) * 1e-9
vals = collections.defaultdict(dict)
names = {}
for r in csv.DictReader(open(cc)):
    vals[r['Dispatch_Id']][r['Counter_Name']] = float(r['Counter_Value'])
    k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    names[r['Dispatch_Id']] = re.sub(r'\(.*$', '', k).replace('void ', '').strip()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
for d, v in vals.items():
    if d not in dur or v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) <= 0:
        continue
    a = agg[names[d]]
    a[0] += 1; a[1] += v['SQ_VALU_MFMA_BUSY_CYCLES']; a[2] += v['SQ_BUSY_CU_CYCLES']; a[3] += dur[d]
    a[4] += v.get('SQ_WAIT_ANY', 0); a[5] += v.get('SQ_WAVE_CYCLES', 0); a[6] += v.get('SQ_LDS_BANK_CONFLICT', 0)
print(f"{'kernel':44s} {'launches':>8s} {'avg_us':>8s} {'mfma_pipe_busy':>15s} {'frac_of_peak (busy x 2.38/2.4 GHz)':>36s} {'wait/wave':>10s} {'lds_conflict':>13s}")
for k, (n, mf, cu, t, wa, wc, lc) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
    busy = (mf / (CUS * 4)) / (cu / CUS)
    print(f'{k[:44]:44s} {n:8d} {t / n * 1e6:8.1f} {busy:15.3f} {busy * STAMPED_CLOCK_GHZ / 2.4:36.3f} {wa / max(wc, 1):10.3f} {lc:13.0f}')
