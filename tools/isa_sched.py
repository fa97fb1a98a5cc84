"""Compact schedule string of the biggest loop of a kernel in an .s file: M mfma, v valu, t transcendental,
a accvgpr move, r/w LDS read/write, g global, s salu, | waitcnt, n nop.  Development aid."""
import re, sys
src, kern = sys.argv[1], sys.argv[2]
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
txt = open(src).read().split('\n')
k0 = [k for k, l in enumerate(txt) if l.startswith(kern + ":")][0]
k1 = [k for k, l in enumerate(txt) if 's_endpgm' in l and k > k0][0]
lines = txt[k0:k1]
labels = {l.split(':')[0]: k for k, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
loops = {}
for k, l in enumerate(lines):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < k and k - labels[m.group(1)] > 300:
        loops[labels[m.group(1)]] = max(k, loops.get(labels[m.group(1)], 0))
loops = sorted(loops.items())
a, b = loops[which]
out = []
for l in lines[a:b + 1]:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'): continue
    x = l.split()[0]
    c = ('M' if 'mfma' in x else 'r' if x.startswith('ds_read') else 'w' if x.startswith('ds_write') else 'd' if x.startswith('ds_')
         else 'g' if x.startswith('global') else 't' if re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)', x) else 'a' if 'accvgpr' in x
         else 'v' if x.startswith('v_') else '|' if x.startswith('s_waitcnt') else 'n' if x.startswith('s_nop') else 's')
    out.append(c)
s = ''.join(out)
print(len(loops), 'loops; showing', which, 'len', len(s), 'M', s.count('M'))
for i in range(0, len(s), 160): print(s[i:i + 160])
