"""Static count of dependent memory round trips per kernel: the number of `s_waitcnt vmcnt` instructions that
have at least one global/buffer load issued since the previous such wait (loops counted once).
    python tools/isa_hops.py [file.hip ...]      (default: every kernel source of the library)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, 'iccv2025-upp_amd/upp_hip/csrc/*.hip')))
    for f in files:
        asm = subprocess.run(['hipcc', '-O3', '--offload-arch=gfx950', '-ffp-contract=off', '-std=c++17', '-S', '--cuda-device-only',
                              '-I' + os.path.join(ROOT, 'include'), f, '-o', '-'], capture_output=True, text=True).stdout
        name, loads, hops, pend, total = None, 0, 0, 0, {}
        for line in asm.splitlines():
            m = re.match(r'^(_Z\w+):', line)
            if m:
                name, loads, hops, pend = m.group(1), 0, 0, 0
                continue
            if name is None:
                continue
            if re.search(r'\b(global_load|buffer_load|flat_load)', line):
                loads += 1; pend += 1
            elif 's_waitcnt' in line and 'vmcnt' in line and pend:
                hops += 1; pend = 0
            elif line.startswith('.Lfunc_end'):
                total[name] = (loads, hops); name = None
        for k, (l, h) in total.items():
            d = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
            d = re.sub(r'\(anonymous namespace\)::', '', d).split('(')[0]
            print('%-14s %-48s loads %4d  wait-after-load %3d' % (os.path.basename(f), d[:48], l, h))


if __name__ == '__main__':
    main()
