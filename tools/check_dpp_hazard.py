#!/usr/bin/env python
"""Static check of the hand-written DPP instructions (wide_kernels.hpp, path_kernels.hpp, gen_kernels.hpp) in the
compiled code.

A DPP read of a VGPR needs two wait states after a VALU write of that VGPR (LLVM
GCNHazardRecognizer::checkDPPHazards, DppVgprWaitStates = 2).  The compiler inserts them for its
own DPP instructions but does not look into inline assembly, where `v_fmac_f64_dpp ...
row_newbcast` lives; the kernels cover the hazard with an `s_nop 1` in the first instruction of
every row group.  This script re-derives that property from the assembly: for every
v_fmac_f64_dpp it walks back until two wait states have passed and fails if a VALU instruction
in that window writes the DPP source register (or if control flow joins inside the window).

    python tools/check_dpp_hazard.py            # compiles wide_api.hip, path_api.hip, gen_api.hip to assembly
    python tools/check_dpp_hazard.py file.s ... # checks given assembly files
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')


def regs(tok):
    m = REG.search(tok)
    if not m:
        return set()
    if m.group(1) is not None:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def check(path):
    lines = open(path).read().split('\n')
    bad, count = [], 0
    for i, ln in enumerate(lines):
        s = ln.strip()
        if not s.startswith('v_fmac_f64_dpp'):
            continue
        count += 1
        ops = s.split(None, 1)[1].split(',')
        src = regs(ops[1])
        waited, j = 0, i - 1
        while waited < 2 and j >= 0:
            t = lines[j].strip()
            j -= 1
            if not t or t.startswith(';') or t.startswith('.') and not t.endswith(':'):
                continue
            if t.endswith(':') or t.split(':')[0].startswith('.LBB') and ':' in t.split()[0]:
                bad.append((i + 1, 'control flow joins within the hazard window', s))
                break
            if t.startswith('s_nop'):
                waited += int(t.split()[1]) + 1
                continue
            op = t.split()[0]
            if op.startswith('v_') and not op.startswith('v_cmp') and not op.startswith('v_readlane'):
                dst = t.split(None, 1)[1].split(',')[0]
                written = regs(dst)
                if op.startswith('v_permlane') and 'swap' in op:   # swaps write both operands
                    written |= regs(t.split(None, 1)[1].split(',')[1])
                if written & src:
                    bad.append((i + 1, 'VALU write of the DPP source %d wait state(s) before' % waited, t))
                    break
            waited += 1
    return count, bad


def main():
    files = sys.argv[1:]
    tmp = None
    if not files:
        tmp = tempfile.mkdtemp(prefix='dpphaz')
        # every translation unit with inline-assembly DPP FMAs: the 64-lane E-step kernels (wide_api), the
        # segment-parallel Viterbi passes for 9..64 states (path_api) and for 65..128 states (gen_api)
        for tu, flags in (('wide_api', []), ('path_api', ['-ffp-contract=off']), ('gen_api', ['-ffp-contract=off'])):
            out = os.path.join(tmp, tu + '.s')
            cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-munsafe-fp-atomics'] + flags + [
                   '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'bhmm_amd', 'csrc'),
                   '--cuda-device-only', '-S', os.path.join(ROOT, 'bhmm_amd', 'csrc', tu + '.hip'), '-o', out]
            subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
            files.append(out)
    total, allbad = 0, []
    for f in files:
        n, bad = check(f)
        total += n
        allbad += [(f,) + b for b in bad]
    print('%d v_fmac_f64_dpp instructions checked, %d hazard(s)' % (total, len(allbad)))
    for b in allbad[:20]:
        print('  %s:%d: %s\n      %s' % b)
    return 1 if allbad or total == 0 else 0


if __name__ == '__main__':
    sys.exit(main())
