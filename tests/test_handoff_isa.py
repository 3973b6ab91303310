"""ISA check of the hand-off polls (csrc/handoff.h::poll_round): in the BUILT code objects of the persistent kernels no
instruction may touch a destination register of a polling `global_load_dwordx4 ... sc1` between that load and the
`s_waitcnt vmcnt(0)` that completes it.  A read there would see the previous poll's data (the intermittent decoder
deviation of round 2); a copy or a spill there would move stale data.  The narrow polls (1, 2, 3, 5 loads)
are one asm statement with early-clobber outputs, where this holds by construction; the wide ones (4, 6..10 loads) keep
separate statements + register pins, because the one-statement form at 10 loads costs 175 spilled registers and 2.4 ms per
training step (csrc/handoff.h) - for them this test IS the guarantee, on the code that ships.  CPU-only: disassembles
lib/obj/*.o with the ROCm llvm-objdump."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'lib', 'obj')
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'

REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def device_isa(name, tmp):
    src = os.path.join(OBJ, name + '.o')
    if not (os.path.exists(src) and os.path.exists(OBJDUMP)):
        pytest.skip('needs the built objects and the ROCm llvm-objdump')
    dst = os.path.join(str(tmp), name + '.o')
    shutil.copy(src, dst)
    subprocess.check_call([OBJDUMP, '--offloading', dst], stdout=subprocess.DEVNULL, cwd=str(tmp))
    dev = [f for f in os.listdir(str(tmp)) if f.startswith(name + '.o.') and 'gfx950' in f]
    assert len(dev) == 1, dev
    return subprocess.check_output([OBJDUMP, '-d', os.path.join(str(tmp), dev[0])], text=True)


def scan(isa):
    """Returns (polling rounds seen, total loads in them, violations)."""
    rounds = loads = 0
    bad = []
    func = '?'
    pending = set()             # destination registers of sc1 polls not yet covered by a vmcnt(0) wait
    for line in isa.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:', line)
        if m:
            assert not pending, (func, 'poll without wait at the end of the function')
            func = m.group(1)
            continue
        ins = line.split('//')[0].strip()
        if not ins:
            continue
        if ins.startswith('global_load_dwordx4') and ' sc1' in ins and ' nt' not in ins:
            ops = ins[len('global_load_dwordx4'):].split(',')
            dst, addr = vregs(ops[0]), vregs(ops[1])
            if (dst | addr) & pending:
                bad.append((func, ins, 'overlaps an outstanding poll destination'))
            if not pending:
                rounds += 1
            loads += 1
            pending |= dst
            continue
        if pending:
            if ins.startswith('s_waitcnt') and 'vmcnt(0)' in ins:
                pending = set()
                continue
            if ins.startswith(('s_branch', 's_cbranch', 's_endpgm', 's_setpc')):
                bad.append((func, ins, 'control flow between a poll and its wait'))
            if vregs(ins) & pending:
                bad.append((func, ins, 'touches v%s before the wait' % sorted(vregs(ins) & pending)))
    return rounds, loads, bad


@pytest.mark.parametrize('name,min_rounds', [('decoder_persist', 4), ('decoder_stream', 4), ('lstm_persist3', 4), ('lstm_persist2', 2)])
def test_no_use_of_poll_destinations_before_the_wait(name, min_rounds, tmp_path):
    rounds, loads, bad = scan(device_isa(name, tmp_path))
    assert rounds >= min_rounds, (name, rounds)          # the check must have seen the polling loops
    assert loads > rounds or name == 'lstm_persist2'
    assert not bad, bad[:10]


def test_scan_flags_a_read_before_the_wait():
    good = """0000000000001000 <k>:
	global_load_dwordx4 v[0:3], v[6:7], off sc1   // 0
	global_load_dwordx4 v[8:11], v[12:13], off sc1
	s_waitcnt vmcnt(0)
	v_and_b32_e32 v20, v0, v21
"""
    assert scan(good) == (1, 2, [])
    bad = good.replace('\ts_waitcnt vmcnt(0)\n\tv_and_b32_e32 v20, v0, v21', '\tv_and_b32_e32 v20, v9, v21\n\ts_waitcnt vmcnt(0)')
    assert len(scan(bad)[2]) == 1
    copy = good.replace('\ts_waitcnt vmcnt(0)', '\tv_mov_b32_e32 v30, v2\n\ts_waitcnt vmcnt(0)')
    assert len(scan(copy)[2]) == 1
