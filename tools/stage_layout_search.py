import re, itertools, sys
src=open('/root/repo/libjxl-tiny_amd/csrc/jxlt_tables.h').read()
m=re.search(r'JXLT_kCoeffOrder\[192\] = \{(.*?)\};',src,re.S)
order=[int(x) for x in re.findall(r'\d+',m.group(1))]
assert len(order)==192
def read32_extra(addrs):   # addrs: list of 64 float addresses (None = inactive); ds_read_b32 rules
    extra=0
    for g in (range(0,32),range(32,64)):
        banks={}
        for ln in g:
            a=addrs[ln]
            if a is None: continue
            banks.setdefault(a%32,set()).add(a)
        extra+=max((len(v) for v in banks.values()),default=1)-1
    return extra
def write128_extra(chunk_addr_of_lane):  # 8 groups of 8 contiguous lanes, banks mod 32, 4 floats each
    extra=0
    for g in range(8):
        banks={}
        for ln in range(8*g,8*g+8):
            a=chunk_addr_of_lane(ln)
            for k in range(4): banks.setdefault((a+k)%32,set()).add(a+k)
        extra+=max(len(v) for v in banks.values())-1
    return extra
def evaluate(sigma, swap, stride):
    # slot of (row r, col l) within a block-channel's 64 floats
    def slot(r,l): return l*8 + (sigma[r] ^ (4 if (swap and l>=4) else 0))
    tot=0
    # class 0: DCT8: n = order[p], r=n>>3,l=n&7
    a=[slot(order[p]>>3, order[p]&7) for p in range(64)]
    tot+=read32_extra(a)   # weight: per DCT8 transform
    c0=tot
    # two-block: n in 0..127: r16 = n>>3 (0..15), l=n&7; second block if r16>=8 (bit 6 of n)
    two=0
    for o2stride in (stride*1, stride*8):
        for cls in (1,2):
            a=[]
            for p in range(64):
                n=order[cls*64+p]
                blk=1 if (n&64) else 0
                a.append(blk*o2stride + slot((n>>3)&7, n&7))
            two+=read32_extra(a)
    # writes: put8: lane l of octet (8 octets per wave = 8 different blocks, block index b -> base b*stride); chunk A (rows with sigma<4) and chunk B
    w=0
    for half in (0,1):
        # octets of a wave store at the same time: lanes 8o..8o+7 -> group o = one octet (8 contiguous lanes): block base irrelevant mod conflicts within the group
        def ca(ln):
            l=ln&7
            return l*8 + ((4*half) ^ (4 if (swap and l>=4) else 0))
        w+=write128_extra(ca)
    return c0, two, w
best=[]
for swap in (0,1):
  for sigma in itertools.permutations(range(8)):
    # rows 0-3 of the chunk must be 4 registers that form chunk A: any; keep all perms
    c0,two,w=evaluate(sigma,swap,200)
    best.append((two*1.0 + c0*0.02 + w*0.5, c0, two, w, swap, sigma))
best.sort()
for b in best[:8]: print(b)
print("current", evaluate(tuple(range(8)),0,200))
for st in (200,196,204,208,216,232):
    print("stride",st,"identity", evaluate(tuple(range(8)),0,st), "swap", evaluate(tuple(range(8)),1,st))
