"""Instruction mix per basic block of one kernel in a `hipcc -S` dump: where are the MFMAs, the scratch spills, the AGPR copies?
    hipcc ... -S --cuda-device-only -o k.s ; python tools/isa_blocks.py k.s <mangled kernel name> [min instructions per block]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]; minn = int(sys.argv[3]) if len(sys.argv) > 3 else 120
on = False; blocks = []; cur = None
for l in txt:
    if l.startswith(name + ':'):
        on = True; cur = {'name': 'entry', 'mfma': 0, 'scr': 0, 'acc': 0, 'valu': 0, 'ds': 0, 'buf': 0, 'salu': 0, 'wait': 0, 'n': 0}; blocks.append(cur); continue
    if not on: continue
    if 's_endpgm' in l: break
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = {'name': m.group(1), 'mfma': 0, 'scr': 0, 'acc': 0, 'valu': 0, 'ds': 0, 'buf': 0, 'salu': 0, 'wait': 0, 'n': 0}; blocks.append(cur); continue
    if not l.startswith('\t') or l.startswith('\t.') or l.startswith('\t;'): continue
    op = l.split()[0]; cur['n'] += 1
    if op.startswith('v_mfma'): cur['mfma'] += 1
    elif op.startswith('scratch_'): cur['scr'] += 1
    elif op.startswith('v_accvgpr'): cur['acc'] += 1
    elif op.startswith('ds_'): cur['ds'] += 1
    elif op.startswith('buffer_') or op.startswith('global_'): cur['buf'] += 1
    elif op.startswith('s_waitcnt'): cur['wait'] += 1
    elif op.startswith('v_'): cur['valu'] += 1
    elif op.startswith('s_'): cur['salu'] += 1
tot = {k: sum(b[k] for b in blocks) for k in blocks[0] if k != 'name'}
print('total', tot)
for b in blocks:
    if b['n'] >= minn: print(b)
