"""tools/isa_stats.py <file.s> <kernel name substring>...: size, registers, scratch, loads, waits and branches of kernels in
hipcc -S output (hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -o out.s file.hip)."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    for m in re.finditer(r'^(_Z\w*' + re.escape(name) + r'\w*):', s, re.M):
        sym = m.group(1)
        a = m.end(); b = s.index('.Lfunc_end', a)
        lines = [l for l in s[a:b].split('\n') if l.strip() and not l.strip().startswith(('.', ';'))]
        mm = re.search(r'\.amdhsa_kernel ' + re.escape(sym) + r'\n(.*?)\.end_amdhsa_kernel', s, re.S).group(1)
        vg = re.search(r'next_free_vgpr (\d+)', mm).group(1); sc = re.search(r'private_segment_fixed_size (\d+)', mm).group(1)
        c = Counter(l.split()[0] for l in lines)
        print(f"{sym[:60]:60s} instrs {len(lines):6d} vgpr {vg:>4s} scratch {sc:>5s} vmem-loads {sum(v for k, v in c.items() if 'load' in k and not k.startswith('s_')):4d} "
              f"s_loads {sum(v for k, v in c.items() if k.startswith('s_load')):4d} waitcnt {c['s_waitcnt']:4d} f64 {sum(v for k, v in c.items() if 'f64' in k):4d} branches {sum(v for k, v in c.items() if k.startswith('s_cbranch')):4d}")
