#!/usr/bin/env python3
"""Per-kernel summary of a gfx950 assembly listing (hipcc -S --cuda-device-only): registers, scratch,
vmcnt waits, barriers, LDS-DMA and global access counts.
    python tools/isa_summary.py listing.s [substring of the kernel name]"""
import re, sys
s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
labels = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\S+): +; @', s, re.M)]
for i, (pos, name) in enumerate(labels):
    end = labels[i + 1][0] if i + 1 < len(labels) else len(s)
    body = s[pos:end]
    if want not in name:
        continue
    code = body.split('.section')[0]
    get = lambda k: (re.search(r'\.amdhsa_%s (\S+)' % k, body) or [None, None])[1]
    print(name[-70:])
    print('   vgpr', get('next_free_vgpr'), 'scratch', get('private_segment_fixed_size'),
          '| glds', code.count('global_load_lds'), 'gstore', len(re.findall(r'global_store_', code)),
          'gload', len(re.findall(r'global_load_dword', code)), 'bufld', code.count('buffer_load'),
          'barriers', code.count('s_barrier'), 'scratch ops', len(re.findall(r'\bscratch_', code)))
    print('   vmcnt waits:', [w.strip() for w in re.findall(r's_waitcnt[^\n]*vmcnt\(\d+\)[^\n]*', code)])
