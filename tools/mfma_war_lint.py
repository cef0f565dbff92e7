#!/usr/bin/env python3
"""Lint of the generated gfx950 ISA for one hazard hipcc (ROCm 7.2) does not cover: a VALU instruction that overwrites the
C operand registers of an fp32 matrix instruction (v_mfma_f32_16x16x4_f32 with vdst != src C, i.e. a renamed result)
fewer than 7 wait states after it.  The compiler inserts s_nop for LDS / memory returns into such registers, and for
VALU writes after the XDL (low-precision) instructions, but not for this pair; on gfx950 the instruction reads C late
when the matrix pipe is contended and the result loses its accumulated sum in the last columns
(csrc/experimental/sepconv_flat.inc: found by a bit-identity test, round 4).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iqpwcnet_amd/csrc -S --cuda-device-only qpwcnet_amd/csrc/optflow.hip -o /tmp/optflow.s
    python tools/mfma_war_lint.py /tmp/optflow.s

Prints every suspect pair.  A suspect is not a proof: the product kernels have a few (distance 0-3: cost volume 4,
encoder ~80, optflow ~160 of which most in f16 kernels, where the compiler's own XDL rule applies) and are run-to-run
stable under tests/test_gpu_determinism.py; a new kernel with suspects must pass that test before it ships."""
import re,sys
S=sys.argv[1]
lines=open(S).read().split('\n')
def rng(tok):
    m=re.match(r'v\[(\d+):(\d+)\]',tok.strip())
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok.strip())
    if m: return {int(m.group(1))}
    return set()
def allregs(tok):
    out=set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b',tok):
        if m.group(1): out|=set(range(int(m.group(1)),int(m.group(2))+1))
        else: out.add(int(m.group(3)))
    return out
ins=[]
for i,l in enumerate(lines):
    c=l.split(';')[0].strip()
    if not c or c.endswith(':') or c.startswith('.'): continue
    ins.append((i+1,c))
n=0
for k,(ln,c) in enumerate(ins):
    if not c.startswith('v_mfma'): continue
    ops=[x.strip() for x in c.split(None,1)[1].split(',')]
    D=rng(ops[0]); C=rng(ops[3]) if len(ops)>3 else set()
    if not C: continue
    ws=0
    for ln2,c2 in ins[k+1:k+12]:
        op=c2.split()[0]
        if op.startswith('s_nop'):
            ws+=int(c2.split()[1])+1; 
        else:
            if op.startswith('v_') and not op.startswith('v_mfma') and not op.startswith('v_cmp'):
                d=allregs(c2.split(None,1)[1].split(',')[0]) if len(c2.split(None,1))>1 else set()
                if d & C and ws<7:
                    print("HAZARD? line %d: %s   then (ws=%d) line %d: %s"%(ln,c,ws,ln2,c2)); n+=1
            # mfma issue: next MFMA waits for pipe (8 passes) -> counts as >= 7 wait states for anything after it
            if op.startswith('v_mfma'): ws+=8
            else: ws+=1
        if ws>=8: break
print("suspects:",n)
