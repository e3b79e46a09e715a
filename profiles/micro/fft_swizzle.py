"""LDS bank conflicts of MfccKernel's in-place split-radix FFT (256 complex points, 64 lanes, 32 banks): passes per frame of every
LDS access of the levels and of the post-processing gather under a layout i -> i ^ g(i >> 5), exhaustive search over the linear g
(5 x 3 bit matrices).  Input: the plan dumped as lines "T level lane kind logm off n" / "P index perm" (a ten-line C++ main over
csrc/srfft_plan.cc prints it).  Result used by engine.cc: columns (14, 21, 1): 636 -> 308 passes per frame, 208 = conflict-free."""
import itertools, collections, sys
tasks=collections.defaultdict(dict); perm={}
for l in open(sys.argv[1] if len(sys.argv) > 1 else 'plan.txt'):
    p=l.split()
    if p[0]=='T': tasks[int(p[1])][int(p[2])]=(int(p[3]),int(p[4]),int(p[5]),int(p[6]))
    else: perm[int(p[1])]=int(p[2])
NC=256
# LDS instructions: list of (list of per-lane index or None) ; each is one b32 access of 64 lanes (xr and xi identical pattern -> weight 2)
instrs=[]   # (weight, [idx per lane])
for L in sorted(tasks):
    # kind 0: 4 points, each read (xr,xi) and written (xr,xi): per point 2 reads + 2 writes = weight 4
    pts=[[None]*64 for _ in range(4)]
    for lane,(kind,lg,off,n) in tasks[L].items():
        if kind==0:
            m=1<<lg; e0=off+n; pts[0][lane]=e0; pts[1][lane]=e0+m//4; pts[2][lane]=e0+m//2; pts[3][lane]=e0+m//2+m//4
        elif kind==1:
            for j in range(4): pts[j][lane]=off+j
        else:
            for j in range(2): pts[j][lane]=off+j
    for j in range(4):
        if any(v is not None for v in pts[j]): instrs.append((4,pts[j],'fft%d'%L))
# post-processing gather: k = lane+1 (+64): reads xr/xi at perm[k] and perm[NC-k]
for it in range(2):
    a=[None]*64; b=[None]*64
    for lane in range(64):
        k=lane+1+64*it
        if 2*k<=NC: a[lane]=perm[k]; b[lane]=perm[NC-k]
    instrs.append((2,a,'post')); instrs.append((2,b,'post'))
def cost(sw):
    tot=0; by=collections.Counter()
    for w,idx,tag in instrs:
        for half in range(2):
            banks=collections.Counter()
            for lane in range(32*half,32*half+32):
                v=idx[lane]
                if v is None: continue
                banks[sw(v)&31]+=1
            c=max(banks.values()) if banks else 0
            tot+=w*c; by[tag]+=w*c
    return tot,by
ident=lambda v:v
t,by=cost(ident); print('identity passes',t,dict(by))
ideal=sum(w*sum(1 for half in range(2) if any(idx[l] is not None for l in range(32*half,32*half+32))) for w,idx,_ in instrs); print('ideal',ideal)
best=None
# linear swizzles: bank bits ^= M * (b5,b6,b7): M columns c5,c6,c7 in 0..31
for c5 in range(32):
  for c6 in range(32):
    for c7 in range(32):
        def sw(v,c5=c5,c6=c6,c7=c7):
            g=(c5 if v&32 else 0)^(c6 if v&64 else 0)^(c7 if v&128 else 0)
            return v^g
        t,_=cost(sw)
        if best is None or t<best[0]: best=(t,c5,c6,c7)
print('best linear',best)
t,c5,c6,c7=best
def sw(v):
    g=(c5 if v&32 else 0)^(c6 if v&64 else 0)^(c7 if v&128 else 0)
    return v^g
print(cost(sw))
