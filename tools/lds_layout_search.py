import itertools
G128=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
      list(range(32,36))+list(range(44,48))+list(range(52,60)), list(range(36,44))+list(range(48,52))+list(range(60,64))]
def cost_read128(addr_of_lane):  # addr in floats (16B aligned), returns extra cycles
    extra=0
    for g in G128:
        banks={}
        for ln in g:
            a=addr_of_lane(ln)
            if a is None: continue
            for k in range(4):
                b=(a+k)%64
                banks.setdefault(b,set()).add(a+k)
        worst=max((len(v) for v in banks.values()), default=1)
        extra+=worst-1
    return extra
def cost_write32(addr_of_lane):
    extra=0
    for g in (range(0,32),range(32,64)):
        banks={}
        for ln in g:
            a=addr_of_lane(ln)
            banks.setdefault(a%32,set()).add(a)
        worst=max(len(v) for v in banks.values())
        extra+=worst-1
    return extra
best=[]
for P in range(64,80,4):
  for A1 in range(32,P-31,4):
    for x0 in range(8):
      for x1 in range(8):
        for ko in range(8):       # octet-dependent row swizzle: row ^ ((o & ko_mask)...) keep simple: row ^ (x ^ (o&ko))
          def pos(o,r,c):
              if c<4: return P*o + 4*((r^x0^(o&ko))&7) + c
              return P*o + A1 + 4*((r^x1^(o&ko))&7) + (c-4)
          w=sum(cost_write32(lambda ln,j=j: pos(ln>>3,j,ln&7)) for j in range(8))
          r=cost_read128(lambda ln: pos(ln>>3,ln&7,0)) + cost_read128(lambda ln: pos(ln>>3,ln&7,4))
          best.append((w*2+r, w, r, P, A1, x0, x1, ko))
best.sort()
for b in best[:12]: print(b)
# current layout
P,A1=72,36
def pos(o,r,c): return P*o + (c>>2)*36 + r*4 + (c&3)
w=sum(cost_write32(lambda ln,j=j: pos(ln>>3,j,ln&7)) for j in range(8))
r=cost_read128(lambda ln: pos(ln>>3,ln&7,0)) + cost_read128(lambda ln: pos(ln>>3,ln&7,4))
print("current", w, r)
print("---- L* (P=64, xor swizzle)")
def posL(o,r,c):
    h=c>>2
    chunk=((r&3)^(o&3)) | ((h^((o>>1)&1))<<2) | ((r>>2)<<3)
    return 64*o + 4*chunk + (c&3)
w=sum(cost_write32(lambda ln,j=j: posL(ln>>3,j,ln&7)) for j in range(8))
r=cost_read128(lambda ln: posL(ln>>3,ln&7,0)) + cost_read128(lambda ln: posL(ln>>3,ln&7,4))
print("L* writes extra", w, "reads extra", r)
# bijection check
for o in range(8):
    assert sorted(posL(o,r,c)-64*o for r in range(8) for c in range(8))==list(range(64))
print("---- pair half areas, pitch 80")
def rd(ln, phase, half):
    o,l=ln>>3,ln&7
    if (l<4) != (phase==0): return None
    return (l&3)*80 + o*8 + 4*half
for ph in (0,1):
    for half in (0,1):
        print("phase",ph,"half",half,"extra",cost_read128(lambda ln: rd(ln,ph,half)))
print("---- pair half areas, current xor layout (pitch 64)")
def rd2(ln, phase, half):
    o,l=ln>>3,ln&7
    if (l<4) != (phase==0): return None
    lr=l&3
    return lr*64 + o*8 + 4*((lr&1)^half)
for ph in (0,1):
    for half in (0,1):
        print("phase",ph,"half",half,"extra",cost_read128(lambda ln: rd2(ln,ph,half)))
