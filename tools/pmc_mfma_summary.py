"""Matrix-pipe occupancy and wave-state breakdown per kernel from one rocprofv3 --pmc pass of tools/prof_kernels.py
(SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES; no trace domain in the
same run) joined with the kernel-trace durations of the separate trace pass.
    python tools/pmc_mfma_summary.py <counter_collection.csv> <kernel_trace.csv> > profiles/r02_pmc_mfma.json
Per kernel (medians over its launches):
  mfma_busy_cycles        SQ_VALU_MFMA_BUSY_CYCLES summed over the chip: cycles in which a SIMD's matrix pipe is busy
                          (64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32)
  mfma_pipe_frac          mfma_busy_cycles / (1024 SIMDs x duration x CLOCK_GHZ) -- share of ALL matrix-pipe cycles of the chip during
                          the launch (launch ramp, idle CUs and tile quantisation included); CLOCK_GHZ = 2.1, the in-kernel clock under
                          this load (tools/micro/lin_stamps.py), so the figure carries that +-5 %
  wait_any / wait_inst / active   SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES (parked on s_waitcnt or
                          a barrier / stalled at issue / issuing)"""
import collections
import csv
import json
import statistics
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from pmc_summary import labelled, name  # noqa: E402

CLOCK_GHZ = 2.1


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    names = sorted({r['Counter_Name'] for r in rows})
    vals = {n: collections.defaultdict(list) for n in names}
    for n in names:
        for k, r in labelled([r for r in rows if r['Counter_Name'] == n], 'Dispatch_Id'):
            vals[n][k].append(float(r['Counter_Value']))
    dur, kern = collections.defaultdict(list), {}
    for k, r in labelled(list(csv.DictReader(open(sys.argv[2]))), 'Dispatch_Id'):
        dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        kern.setdefault(k, name(r))
    out = {}
    med = lambda n, k: statistics.median(vals[n][k]) if n in vals and k in vals[n] else None
    for k in sorted(vals.get('SQ_VALU_MFMA_BUSY_CYCLES', {})):
        if k.startswith(('at::', '__amd', 'Cijk', 'rocprim', 'elementwise', 'randperm', 'softmax_warp')):
            continue
        mf, wc = med('SQ_VALU_MFMA_BUSY_CYCLES', k), med('SQ_WAVE_CYCLES', k)
        if not mf:
            continue
        us = statistics.median(dur[k]) if k in dur else None
        e = {'us': us, 'kernel': kern.get(k), 'mfma_busy_cycles': mf, 'mfma_pipe_frac': mf / (1024 * us * 1e3 * CLOCK_GHZ) if us else None,
             'clock_ghz': CLOCK_GHZ}
        for n, label in (('SQ_WAIT_ANY', 'wait_any'), ('SQ_WAIT_INST_ANY', 'wait_inst'), ('SQ_ACTIVE_INST_ANY', 'active')):
            v = med(n, k)
            e[label] = v / wc if (v is not None and wc) else None
        out[k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
