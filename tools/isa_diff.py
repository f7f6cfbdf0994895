import re,sys
def funcs(path):
    out={}; cur=None
    for line in open(path):
        l=line.split(';')[0].rstrip()
        s=l.strip()
        if not s or s.startswith('.file') or s.startswith('.loc'): continue
        m=re.match(r'^(_Z\w+):',l)
        if m:
            cur=m.group(1); out[cur]=[]; continue
        if s.startswith('.end_amdhsa_kernel') or l.startswith('.Lfunc_end') or s.startswith('.section'):
            cur=None
        if cur:
            if re.match(r's_load_dword\w* .*0x1[89a-f][0-9a-f]$',s) or re.match(r's_load_dword\w* .*0x2[0-9a-f][0-9a-f]$',s): s='HIDDEN_ARG_LOAD'
            if s.startswith('.amdhsa_kernarg_size'): continue
            out[cur].append(re.sub(r'__hip_cuid_\w+','CUID',s))
    return out
for f in sys.argv[1:]:  # usage: isa_diff.py <name> ... compares /tmp/isa/base/<name>.s with /tmp/isa/new/<name>.s
    a=funcs('/tmp/isa/base/%s.s'%f); b=funcs('/tmp/isa/new/%s.s'%f)
    nd=0
    for k in a:
        if a[k]!=b.get(k):
            nd+=1
            d=sum(1 for x,y in zip(a[k],b.get(k,[])) if x!=y)
            print('  DIFF',f,k[:90],len(a[k]),len(b.get(k,[])),'lines differing',d)
    print(f,'kernels',len(a),'different',nd, 'missing/new', set(a)^set(b))
