#!/usr/bin/env python3
"""Bank-conflict checker / layout search for the 8-byte LDS exchange images of fmc_bluestein.h (see tools/README.md)."""
# LDS bank-conflict checker for 8-byte accesses (MI355X_MICROARCH.md LDS table)
#  ds_read_b64 : lane groups {0-31},{32-63}; bank of dword d = d mod 64 ; each lane touches dwords 2e, 2e+1
#  ds_write_b64: 4 groups of 16 contiguous lanes; bank = dword mod 32
import itertools
def cycles(addrs, kind):
    # addrs: list of 64 element indices (8-byte elements) or None for inactive
    groups = [range(0,32), range(32,64)] if kind=="r" else [range(16*g,16*g+16) for g in range(4)]
    nb = 64 if kind=="r" else 32
    tot=0
    for g in groups:
        per_bank={}
        for l in g:
            e=addrs[l]
            if e is None: continue
            for d in (2*e, 2*e+1):
                per_bank.setdefault(d%nb,set()).add(d)
        tot+=max([len(v) for v in per_bank.values()] or [0])
    return tot
def check_full2b(P, FL, FB):
    worst_r=worst_w=0
    NB=(P+7)//8
    # writes: lane (l0=lane&7, i=lane>>3), per (jj,b0): addr=(i+8jj)+FL*l0+FB*b0
    for jj in range(NB):
        for b0 in range(8):
            a=[((l>>3)+8*jj)+FL*(l&7)+FB*b0 if (l>>3)+8*jj<P else None for l in range(64)]
            worst_w=max(worst_w,cycles(a,"w"))
    # full reads: lane (i=lane&7 -> a=i+8jj, b0=lane>>3), per (jj,l0)
    for jj in range(NB):
        for l0 in range(8):
            a=[((l&7)+8*jj)+FL*l0+FB*(l>>3) if (l&7)+8*jj<P else None for l in range(64)]
            worst_r=max(worst_r,cycles(a,"r"))
    return worst_w, worst_r
for P in (4,8,16,24,32):
    FL=P+2
    best=None
    for FB in range((P-1)+FL*7+1, (P-1)+FL*7+1+80):
        w,r=check_full2b(P,FL,FB)
        if w<=4 and r<=2:
            best=FB;break
    print("P",P,"FL",FL,"FB",best, check_full2b(P,FL,best) if best else None, "XELEMS", 8*best if best else None, "old", max(P*72, 8*8*FL))
print("relayout")
def check_relayout(P, adr):
    NB=(P+7)//8
    ww=rr=0
    for jj in range(NB):
        for b1 in range(8):
            a=[adr(((l&7)+8*jj)+P*((l>>3)+8*b1)) if (l&7)+8*jj<P else None for l in range(64)]
            ww=max(ww,cycles(a,"w"))
    for j in range(P):
        a=[adr(l+64*j) for l in range(64)]
        rr=max(rr,cycles(a,"r"))
    return ww,rr
for P in (4,8,16,24,32):
    found=None
    for W in (8,16,32,64,P,2*P,4*P,8*P):
        for pad in range(0,9):
            adr=lambda x,W=W,pad=pad: x+pad*(x//W)
            w,r=check_relayout(P,adr)
            if w<=4 and r<=2:
                size=adr(64*P-1)+1
                if found is None or size<found[2]: found=(W,pad,size)
    print("P",P,found)
print("relayout2")
def check_relayout2(P, adr, mapping):
    NB=(P+7)//8
    ww=rr=0
    for jj in range(NB):
        for b1 in range(8):
            a=[]
            for l in range(64):
                i,b0 = ((l&7),(l>>3)) if mapping==0 else ((l>>3),(l&7))
                a.append(adr((i+8*jj)+P*(b0+8*b1)) if i+8*jj<P else None)
            ww=max(ww,cycles(a,"w"))
    for j in range(P):
        a=[adr(l+64*j) for l in range(64)]
        rr=max(rr,cycles(a,"r"))
    return ww,rr
P=16
res=[]
for mapping in (0,1):
  for W1 in (8,16,32):
    for p1 in range(0,33):
      for W2 in (64,128,256):
        for p2 in range(0,17):
            adr=lambda x: x+p1*(x//W1)+p2*(x//W2)
            w,r=check_relayout2(P,adr,mapping)
            if w<=4 and r<=2:
                res.append((adr(64*P-1)+1,mapping,W1,p1,W2,p2))
res.sort(); print(res[:5])
# xor swizzles
res=[]
for mapping in (0,1):
  for sh in range(1,8):
    for mask in (1,3,7,15):
      for mul in (1,2,4,8,16):
        adr=lambda x: x ^ (((x>>sh)&mask)*mul)
        w,r=check_relayout2(P,adr,mapping)
        if w<=4 and r<=2: res.append((mapping,sh,mask,mul))
print(res[:10])
print("relayout3")
for P in (4,8,16,24,32):
    res=[]
    for mapping in (0,1):
      for sh in range(1,10):
        for mask in (1,3,7,15,31):
          for mul in (1,2,4,8,16,32):
            adr=lambda x: x ^ (((x>>sh)&mask)*mul)
            vals=[adr(x) for x in range(64*P)]
            if len(set(vals))!=64*P: continue
            w,r=check_relayout2(P,adr,mapping)
            if w<=4 and r<=2: res.append((max(vals)+1,mapping,sh,mask,mul))
    res.sort()
    print(P,res[:6])
print("more P")
for P in (10,12,20):
    FL=P+2
    best=None
    for FB in range((P-1)+FL*7+1, (P-1)+FL*7+1+120):
        # reader mapping i=lane&7,b0=lane>>3
        w,r=check_full2b(P,FL,FB)
        if w<=4 and r<=2: best=FB;break
    print("P",P,"FL",FL,"FB",best, "XELEMS", 8*best if best else None, "old", max(P*72, 8*8*FL))
    res=[]
    for sh in range(1,10):
        for mask in (0,1,3,7,15,31):
          for mul in (1,2,4,8,16,32):
            adr=lambda x: x ^ (((x>>sh)&mask)*mul)
            vals=[adr(x) for x in range(64*P)]
            if len(set(vals))!=64*P: continue
            w,r=check_relayout2(P,adr,0)
            if w<=4 and r<=2: res.append((max(vals)+1,sh,mask,mul))
    res.sort(); print(P,res[:4])
print("50-lane family (fmc_mrfft.h): exchange 1 at 69 a + l, exchange 2 dense at q + 5 P b0 (q = 5 a + l0); worst cycles per access"
      " (ideal: write 4 = one per 16-lane group, read 2 = one per 32-lane group)")
for P in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 24):
    L0, L1, SE, NBF = 5, 10, 69, 5 * P
    NB = (NBF + 63) // 64
    w1 = max(cycles([a * SE + l for l in range(64)], "w") for a in range(P))
    r1 = 0
    for jj in range(NB):
        for m in range(L1):
            r1 = max(r1, cycles([((l + 64 * jj) // L0) * SE + (l + 64 * jj) % L0 + L0 * m if l + 64 * jj < NBF else None
                                 for l in range(64)], "r"))
    w2 = 0
    for jj in range(NB):
        for b0 in range(L1):
            w2 = max(w2, cycles([l + 64 * jj + NBF * b0 if l + 64 * jj < NBF else None for l in range(64)], "w"))
    r2 = 0
    N = 50 * P
    for lo in range(0, N - 63):          # every window position, first slot (64 consecutive outputs)
        for m in range(L0):
            r2 = max(r2, cycles([L0 * ((lo + l) % (P * L1)) + m for l in range(64)], "r"))
    print("P", P, "exchange 1 write/read", w1, r1, " exchange 2 write/read", w2, r2)


# ---------------------------------------------------------------- 16-byte images: the one-pass exchange 2 of the float64 16 x 4 row
# Lane groups and banking per instruction (MI355X_MICROARCH.md, LDS table):
#   ds_read_b128 : 4 groups {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59} {36-43,48-51,60-63}, bank = dword mod 64
#   ds_write_b128: 8 groups of 8 contiguous lanes, bank = dword mod 32
#   ds_write2_b64: two accesses (the two 8-byte halves), each 4 groups of 16 contiguous lanes, bank = dword mod 32
#                  (what hipcc emits for a 16-byte store it only knows to be 8-byte aligned)
R128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def cycles_dwords(lane_dwords, groups, nbanks):
    """lane_dwords[l]: the dwords lane l touches in ONE access (None: inactive); LDS-array cycles over the lane groups."""
    tot = 0
    for g in groups:
        per_bank = {}
        for l in g:
            if lane_dwords[l] is None:
                continue
            for d in lane_dwords[l]:
                per_bank.setdefault(d % nbanks, set()).add(d)
        tot += max([len(v) for v in per_bank.values()] or [0])
    return tot


def read_b128(elems):       # elems[l]: index of the 16-byte element lane l reads
    return cycles_dwords([None if e is None else range(4 * e, 4 * e + 4) for e in elems], R128, 64)


def write_b128(elems):
    return cycles_dwords([None if e is None else range(4 * e, 4 * e + 4) for e in elems], [range(8 * g, 8 * g + 8) for g in range(8)], 32)


def write2_b64(elems):      # the same 16-byte store as two 8-byte accesses
    g16 = [range(16 * g, 16 * g + 16) for g in range(4)]
    return sum(cycles_dwords([None if e is None else range(4 * e + 2 * half, 4 * e + 2 * half + 2) for e in elems], g16, 32) for half in (0, 1))


print("16 x 4 row at P = 16 (fmc_wavefft.h: pruned_row_fft_d16r), float64: exchange 1 (8-byte image 66 a + l, reads 66 a + l0 + 4 l1)")
w = max(cycles([a * 66 + l for l in range(64)], "w") for a in range(16))
r = max(cycles([(l & 15) * 66 + (l >> 4) + 4 * l1 for l in range(64)], "r") for l1 in range(16))
print("  worst write", w, "(ideal 4)  worst read", r, "(ideal 2)")
for NP in (6, 8):
    st128 = max(write_b128([(l & 15) + 16 * p + 16 * NP * (l >> 4) for l in range(64)]) for p in range(NP))
    st2 = max(write2_b64([(l & 15) + 16 * p + 16 * NP * (l >> 4) for l in range(64)]) for p in range(NP))
    worst = 0
    for lo in range(0, 1024 - 63):        # every window position, 64 consecutive outputs, the four terms m
        for m in range(4):
            worst = max(worst, read_b128([((lo + l + 8 * NP) & 255) % (16 * NP) + 16 * NP * m for l in range(64)]))
    print(f"  one-pass exchange 2, {NP} planes of 16-byte elements: store as ds_write_b128 {st128} cycles (ideal 8), as ds_write2_b64 {st2} "
          f"(ideal 8: the 2-way conflict of round 2), ds_read_b128 worst over all windows {worst} (ideal 4)")

# ---- packed rows (fmc_wavefft.h: packed_row_fft): exchange-1 image SE a + lane, read inside the groups of L lanes
print("packed rows: exchange-1 reads (ideal 2 cycles) / writes (ideal 4) by image stride")
for L, name in ((8, "N=128"), (16, "N=256"), (32, "N=512")):
    for SE in (64, 65, 66, 67):
        worst = 0
        if L == 8:
            pats = [[((l & 7) + 8 * m) * SE + (l - (l & 7)) + qq for l in range(64)] for m in range(2) for qq in range(8)]
        else:
            L0 = L // 16
            pats = [[((l & (L - 1)) & 15) * SE + (l - (l & (L - 1))) + ((l & (L - 1)) >> 4) + L0 * l1 for l in range(64)] for l1 in range(16)]
        for a in pats:
            worst = max(worst, cycles(a, "r"))
        ww = max(cycles([aa * SE + l for l in range(64)], "w") for aa in range(16))
        print(" ", name, "SE", SE, "read", worst, "write", ww)
