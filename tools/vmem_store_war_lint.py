#!/usr/bin/env python3
"""Lint of generated gfx950 ISA for a second hazard hipcc (ROCm 7.2) leaves open: a buffer store of more than 8 bytes whose
SCALAR OFFSET IS A REGISTER, followed at once by a vector instruction that overwrites the store's data registers.  The
compiler's rule (GCNHazardRecognizer, "12-dword store data hazard") inserts the wait state only when the scalar offset is
an inline constant; with an SGPR there it assumes the data has been read.  On gfx950 it has not: the fp32 SeparableConv2D
epilogue returned garbage in one channel of one block of 16 outputs, different from run to run, the first time its block
offset (64 * ft bytes) went into the scalar offset (round 4).  Write block offsets into the address register or the
immediate field instead.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iqpwcnet_amd/csrc -Iinclude -S --cuda-device-only qpwcnet_amd/csrc/optflow.hip -o /tmp/optflow.s
    python tools/vmem_store_war_lint.py /tmp/optflow.s        # prints every suspect; exit status 1 if any
"""
import re
import sys


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def main(path):
    ins = []
    func = "?"
    for i, l in enumerate(open(path).read().split('\n')):
        c = l.split(';')[0].strip()
        m = re.match(r'^(_Z\w+):', c)
        if m:
            func = m.group(1)
            continue
        if not c or c.endswith(':') or c.startswith('.'):
            continue
        ins.append((i + 1, c, func))
    n = 0
    for k, (ln, c, f) in enumerate(ins):
        m = re.match(r'(buffer_store_dwordx[34]|buffer_store_b(?:96|128))\s+(v\[\d+:\d+\]),\s*([^,]+),\s*(s\[\d+:\d+\]),\s*(\S+)', c)
        if not m:
            continue
        soff = m.group(5).rstrip(',')
        if not re.match(r's\d+$|s\[\d+', soff) and soff not in ("m0",):
            continue                                   # inline constant / literal: the compiler's own rule applies
        data = regs(m.group(2))
        ws = 0
        for ln2, c2, _ in ins[k + 1:k + 4]:
            op = c2.split()[0]
            if op.startswith('s_nop'):
                ws += int(c2.split()[1]) + 1
                continue
            if op.startswith('v_') and not op.startswith('v_cmp') and len(c2.split(None, 1)) > 1:
                d = regs(c2.split(None, 1)[1].split(',')[0])
                if d & data and ws < 2:
                    print("STORE-DATA WAR? %s\n   line %d: %s\n   line %d (wait states %d): %s" % (f, ln, c, ln2, ws, c2))
                    n += 1
            ws += 1
            if ws >= 2:
                break
    print("suspects:", n)
    return 1 if n else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
