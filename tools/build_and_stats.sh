#!/bin/bash
# build librsa_hip.so and print register use + main-loop VALU histogram of one kernel source (default: fp8 K5)
C=/root/repo/rectified_spaattn_amd/csrc
make -C $C 2>&1 | grep -E "error|warning" -A4 | head -20
SRC=${1:-rsa_attn_fp8_kernel.hip}
(cd $C && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I. -fno-honor-nans --cuda-device-only -S -o /tmp/k.s $SRC 2>&1 | grep -v "warning\|hip-link" | head -5)
grep -E "\.vgpr_count|vgpr_spill" /tmp/k.s | head -4
python3 - <<'PY'
import re,collections
lines=open('/tmp/k.s').read().split('\n')
blocks=[];cur=None
for l in lines:
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m:
        cur=[m.group(1),[]];blocks.append(cur);continue
    if cur is None: continue
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur[1].append(t)
for name,ins in blocks:
    mf=sum(1 for i in ins if 'mfma' in i)
    if mf>=8:
        c=collections.Counter(i.split()[0] for i in ins)
        print(name,len(ins),{k:v for k,v in c.items() if k.startswith('v_')}, 'scratch', sum(v for k,v in c.items() if k.startswith('scratch')))
PY
