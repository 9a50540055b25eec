#!/usr/bin/env python3
"""Instruction mix of one kernel in a device-only assembly listing (hipcc --cuda-device-only -S): total, per unit (v_ / s_ / ds_ /
buffer_ / global_), the most frequent opcodes, VGPRs / SGPRs / LDS from the kernel descriptor.  Usage: isa_count.py file.s <substring of the mangled name>"""
import re
import sys
from collections import Counter


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2]
    for m in re.finditer(r'^(\S*%s\S*):[^\n]*\n' % re.escape(want), s, re.M):
        name = m.group(1)
        try:
            b = s.index('.amdhsa_kernel ' + name)
        except ValueError:
            continue
        body = s[m.end():b]
        body = body[:body.index('s_endpgm')] if 's_endpgm' in body else body
        ins = []
        for l in body.split('\n'):
            t = l.strip()
            if not t or t.startswith(('.', ';')) or t.endswith(':'):
                continue
            ins.append(t.split()[0])
        c = Counter(i.split('_')[0] for i in ins)
        d = s[b:s.index('.end_amdhsa_kernel', b)]
        res = {k: re.search(k + r'\s+(\S+)', d).group(1) for k in ('next_free_vgpr', 'next_free_sgpr', 'group_segment_fixed_size', 'private_segment_fixed_size')}
        print(name)
        print("  instructions %d  %s  %s" % (len(ins), dict(c), res))
        print("  " + ", ".join("%s %d" % kv for kv in sorted(Counter(ins).items(), key=lambda kv: -kv[1])[:28]))


if __name__ == "__main__":
    main()
