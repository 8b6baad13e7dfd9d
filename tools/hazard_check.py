#!/usr/bin/env python3
"""tools/hazard_check.py [file.s] -- static check of the inline-asm selects in the decode kernels.

gfx940+ needs two wait states between a VALU instruction that writes an SGPR (v_cmp ... s[a:b]) and a
VALU instruction that reads it; the hardware does not interlock and LLVM's hazard recogniser pads
compiler-generated pairs with s_nop -- but it does not look into inline asm.  csrc/mlp_decode.h places
a few v_cndmask_b32_e64 selects as inline asm precisely so that their compares can be issued well
ahead (no padding).  This script compiles csrc/mlp_hip.hip to assembly (or reads the given .s) and
verifies, for every such select in every kernel, that at least two wait states separate it from the
definition of its mask in the same basic block.  tests/test_cabi.py runs it on every build."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(path):
    L = open(path).read().split('\n')
    viol, checked, unknown = [], 0, 0
    for i, l in enumerate(L):
        t = l.strip()
        if not t.startswith('v_cndmask_b32_e64'):
            continue
        if 'ASMSTART' not in L[i - 1] and not (L[i - 1].strip().startswith('s_nop') and 'ASMSTART' in L[i - 2]):
            continue                                   # compiler-generated: padded by the compiler
        m = re.search(r'(s\[\d+:\d+\])\s*$', t)
        if not m:
            continue
        sg, ws, k, found = m.group(1), 0, i - 1, None
        while k >= 0:
            p = L[k].strip()
            if not p or p.startswith(';') or (p.startswith('.') and not p.endswith(':')):
                k -= 1
                continue
            if p.endswith(':'):
                found = ('label', p)
                break
            mm = re.match(r'^(\S+)\s+(\S+?),', p)
            if mm and mm.group(2) == sg:
                found = ('def', p)
                break
            n = re.match(r'^s_nop\s+(\d+)', p)
            ws += (int(n.group(1)) + 1) if n else 1
            k -= 1
        checked += 1
        if found is None or found[0] == 'label':
            unknown += 1
        elif found[1].startswith('v_') and ws < 2:
            viol.append((i + 1, ws, found[1], t))
    # asm PCM stores (global_store_dwordx4 ..., v[x:y], off ...): a VMEM store of more than 64 bits may
    # still be reading its data registers when the next instructions issue -- a VALU write of them needs
    # two wait states on gfx940+ (LLVM GCNHazardRecognizer::createsVALUHazard, "store data over written
    # by the next instruction"); the compiler pads its own stores, not the ones inside inline asm
    for i, l in enumerate(L):
        t = l.strip()
        if not t.startswith('global_store_dwordx4') or not any('ASMSTART' in L[j] for j in (i - 1, i - 2)):
            continue
        m = re.search(r'v\[(\d+):(\d+)\], off', t)
        if not m:
            continue
        lo, hi = int(m.group(1)), int(m.group(2))
        checked += 1
        ws, k = 0, i + 1
        while k < len(L) and ws < 2:
            p = L[k].strip()
            k += 1
            if not p or p.startswith(';') or (p.startswith('.') and not p.endswith(':')):
                continue
            if p.endswith(':') or p.startswith('s_cbranch') or p.startswith('s_branch') or p.startswith('s_endpgm'):
                break                                  # control flow: at least one more issue cycle, keep it simple
            n = re.match(r'^s_nop\s+(\d+)', p)
            if n:
                ws += int(n.group(1)) + 1
                continue
            mm = re.match(r'^(v_\S+)\s+v\[?(\d+)(?::(\d+))?\]?', p)
            if mm and not mm.group(1).startswith('v_cmp'):
                a = int(mm.group(2))
                b = int(mm.group(3)) if mm.group(3) else a
                if not (b < lo or a > hi):
                    viol.append((k, ws, p, t))
            ws += 1
    # (round 5) the row loop's window read placed as inline asm (ds_read2st64_b32 into the window's own register pair)
    # and waited for by an asm s_waitcnt at the next slot's top: the compiler does not know that the pair's data is
    # still on its way in between, so NOTHING there may read or write the pair -- a register copy at the join of the
    # lanes that carry a slot and those that do not did once (wrong PCM on fuzz streams).  Followed in layout order
    # up to the next asm wait (labels and branches are passed through: the code in between is straight fall-through
    # plus the exec-mask bookkeeping of the `if`).
    i = 0
    while i < len(L):
        t = L[i].strip()
        if t.startswith('ds_read2st64_b32') and 'ASMSTART' in L[i - 1]:
            m = re.search(r'v\[(\d+):(\d+)\]', t)
            lo, hi = int(m.group(1)), int(m.group(2))
            checked += 1
            k, found = i + 1, False
            while k < len(L) and k < i + 400:
                p = L[k].strip()
                k += 1
                if not p or p.startswith(';') or (p.startswith('.') and not p.endswith(':')) or p.endswith(':'):
                    continue
                if p.startswith('s_waitcnt') and 'lgkmcnt(0)' in p:
                    found = True
                    break
                if p.startswith('s_endpgm') or p.startswith('s_setpc'):
                    break
                regs = set()
                for mm in re.finditer(r'\bv\[(\d+):(\d+)\]', p):
                    regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
                for mm in re.finditer(r'\bv(\d+)\b', p):
                    regs.add(int(mm.group(1)))
                if any(lo <= r <= hi for r in regs):
                    viol.append((k, 0, p, t))
                    break
            if not found and not (viol and viol[-1][3] == t):
                unknown += 1
        i += 1
    return checked, unknown, viol


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
        hipcc = "/opt/rocm/bin/hipcc"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                        os.path.join(ROOT, "libdvd-audio_amd", "csrc", "mlp_hip.hip")], check=True,
                       stderr=subprocess.DEVNULL)
        path = out
    checked, unknown, viol = check(path)
    print("inline-asm selects + stores: %d, mask defined in another block: %d, hazard violations: %d" % (checked, unknown, len(viol)))
    for v in viol[:10]:
        print("  line %d: only %d wait state(s) between `%s` and `%s`" % v)
    return 1 if (viol or unknown or checked == 0) else 0


if __name__ == "__main__":
    sys.exit(main())
