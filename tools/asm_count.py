#!/usr/bin/env python3
"""Static instruction census of the kernels in a hipcc -S listing (no GPU needed).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -S --cuda-device-only X.hip -o X.s
    tools/asm_count.py X.s 'gg_pl_kernel<3, 128, false, false, false, false>'

Per kernel: instructions by class over the whole body and over the part after the last MFMA (the epilogue of a GEMM
kernel; it holds several mutually exclusive branches, so the figure is an upper bound of what one wave executes).
"""
import re
import subprocess
import sys
from collections import Counter


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return 'mfma'
    if op.startswith('v_pk_'):
        return 'valu_pk'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt') or op.startswith('s_barrier') or op.startswith('s_nop') or op.startswith('s_sleep'):
        return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('buffer_') or op.startswith('global_') or op.startswith('flat_') or op.startswith('scratch_'):
        return 'vmem'
    return 'other'


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    text = open(path).read().split('\n')
    kernels = {}
    cur = None
    for line in text:
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is None:
            continue
        if line.startswith('.Lfunc_end') or line.strip().startswith('.section'):
            cur = None
            continue
        s = line.strip()
        if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
            continue
        kernels[cur].append(s.split()[0])
    names = subprocess.run(['c++filt'], input='\n'.join(kernels), capture_output=True, text=True).stdout.split('\n')
    for mangled, name in zip(kernels, names):
        if filt not in name:
            continue
        ops = kernels[mangled]
        last = max((i for i, o in enumerate(ops) if o.startswith('v_mfma')), default=-1)
        whole = Counter(classify(o) for o in ops)
        tail = Counter(classify(o) for o in ops[last + 1:])
        print(name)
        print('   whole     :', dict(sorted(whole.items())))
        print('   after MFMA:', dict(sorted(tail.items())))
        if '--top' in sys.argv:
            print('   top tail ops:', Counter(ops[last + 1:]).most_common(25))


if __name__ == '__main__':
    main()
