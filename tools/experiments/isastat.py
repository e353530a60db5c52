import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s=open('build/asm/c2.s').read().split('\n')
idx=[i for i,l in enumerate(s) if 'v_mfma' in l]
F,L=idx[0],idx[-1]
cnt=0; split=L
for i in range(F,L+1):
    if 'v_mfma' in s[i]:
        cnt+=1
        if cnt==820: split=i; break
def stats(a,b):
    body=s[a:b+1]
    bad=sum(1 for i in range(1,len(body)) if 's_waitcnt lgkmcnt(0)' in body[i] and 'ds_read' in body[i-1])
    w=sum(1 for l in body if 's_waitcnt' in l)
    m=sum(1 for l in body if 'v_mfma' in l)
    n=sum(1 for l in body if l.strip().startswith(('v_','s_','ds_','global_')))
    acc=sum(1 for l in body if 'v_accvgpr' in l)
    nops=sum(int(l.split()[1])+1 for l in body if 's_nop' in l)
    print('mfma',m,'instr',n,'waits',w,'bad',bad,'accvgpr',acc,'nop cycles',nops)
stats(F,split); stats(split+1,L)
