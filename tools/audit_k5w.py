import re,sys
lines=open(sys.argv[1]).read().split('\n')
kname=sys.argv[2] if len(sys.argv)>2 else '_Z14bsfwd64_kernelI8bf16_tagLb1ELi0EEv8AttnArgs'
start=[i for i,l in enumerate(lines) if l.startswith(kname+':')][0]
end=[i for i,l in enumerate(lines) if i>start and l.strip().startswith('.end_amdhsa_kernel')][0]
inasm=False; cnt={}; asm_idx=0; asm_first={}
for i in range(start,end):
    l=lines[i].strip()
    if l.startswith(';;#ASMSTART'): inasm=True; asm_idx+=1; asm_first[asm_idx]=lines[i+1].strip()[:50]; continue
    if l.startswith(';;#ASMEND'): inasm=False; continue
    if not inasm:
        m=re.match(r'(v_accvgpr_\w+|v_writelane_b32|v_readlane_b32|scratch_\w+)',l)
        if m: cnt.setdefault((asm_idx,m.group(1)),0); cnt[(asm_idx,m.group(1))]+=1
tot={}
for (a,k),v in cnt.items(): tot[k]=tot.get(k,0)+v
print(tot, 'n asm', asm_idx, 'lines', end-start)
per={}
for (a,k),v in cnt.items(): per.setdefault(a,{})[k]=v
for a in sorted(per): print(a, per[a], '| after asm starting:', asm_first.get(a,''))
