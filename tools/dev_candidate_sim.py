import random, collections
BITS=62; MASK=(1<<BITS)-1
def rotl(v,s):
    s%=BITS
    return ((v<<s)&MASK)|(v>>(BITS-s)) if s else v
def cands(x, nref):
    if x==0 or x==MASK: return [1]*(nref+1)
    z=~x&MASK
    r=z; L=1
    while True:
        t=r&rotl(r,1)
        if t==0: break
        r=t; L+=1
    out=[bin(r).count('1')]
    for j in range(nref):
        t=r&rotl(z,L+1+j)
        if t: r=t
        out.append(bin(r).count('1'))
    return out
random.seed(1)
NREF=3
# reads of 150 random bases -> 120 k-mers; waves of 64 consecutive k-mers (q..q+63) in stream
hist=[collections.Counter() for _ in range(NREF+1)]
wavemax=[collections.Counter() for _ in range(NREF+1)]
stream=[]
for rd in range(600):
    bases=[random.randrange(4) for _ in range(150)]
    x=0
    for i,b in enumerate(bases):
        x=((x<<2)|b)&MASK
        if i>=30: stream.append(x)
cs=[cands(x,NREF) for x in stream]
for c in cs:
    for j in range(NREF+1): hist[j][min(c[j],5)]+=1
for w in range(0,len(cs)-63,64):
    for j in range(NREF+1):
        wavemax[j][min(max(c[j] for c in cs[w:w+64]),6)]+=1
n=len(cs); nw=sum(wavemax[0].values())
for j in range(NREF+1):
    print("refines",j,"lane dist",{k:round(v/n,4) for k,v in sorted(hist[j].items())},"E wave max",round(sum(k*v for k,v in wavemax[j].items())/nw,2),{k:round(v/nw,3) for k,v in sorted(wavemax[j].items())})
    # pairs per iteration
    print("   E wave ceil(max/2)",round(sum(((k+1)//2)*v for k,v in wavemax[j].items())/nw,2))
