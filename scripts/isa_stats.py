"""Per-kernel ISA statistics of a hipcc -save-temps .s file (VGPRs, LDS, scalar / vector loads, waits): a quick check that
a refactoring left a kernel's code shape alone.   usage: python3 scripts/isa_stats.py file.s [name-substring]"""
import re
import sys

s = open(sys.argv[1]).read()
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\S+):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if filt not in name:
        continue
    g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\d+)', body) or [0, "?"])[1]
    print(name[:90].ljust(92), 'vgpr', g('next_free_vgpr'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'),
          's_load', len(re.findall(r'\ts_load', body)), 'gload', len(re.findall(r'\tglobal_load', body)), 'gstore', len(re.findall(r'\tglobal_store', body)),
          'ds', len(re.findall(r'\tds_', body)), 'waitcnt', len(re.findall(r's_waitcnt', body)), 'lines', body.count('\n'))
