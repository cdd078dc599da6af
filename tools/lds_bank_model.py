import itertools
groups=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
groups += [[l+32 for l in g] for g in groups]
def cost(addr_dw):  # addr_dw: list of 64 dword addresses (16B aligned → multiple of 4)
    tot=0
    for grp in groups:
        banks={}
        for l in grp:
            a=addr_dw[l]
            for d in range(4):
                banks.setdefault((a+d)%64,set()).add(a+d)
        tot+=max(len(v) for v in banks.values())
    return tot  # 4 = conflict-free
def pix_cost(P,C=48,HWp=18,perm=None):
    U=C//8; KG=9*U; KS=(KG+3)//4
    res=[]
    for ks in range(KS):
        addr=[]
        for l in range(64):
            g=l//16; j=l%16
            kg=ks*4+g
            if kg>=KG: addr.append(10**6*4); continue
            tap=kg//U; cg=kg%U; dy=tap//3; dx=tap%3
            addr.append((j+dy*HWp+dx)*P+cg*4)
        res.append(cost(addr))
    return res
def w_cost(WPd):
    addr=[(l%16)*WPd+(l//16)*4 for l in range(64)]
    return cost(addr)
for P in range(24,68,4):
    r=pix_cost(P); print("pix pitch dw",P,"sum",sum(r),r)
for WPd in range(224,260,4):
    print("w pitch dw",WPd,w_cost(WPd))

# ---- the GEMM engines' K-contiguous images (gemm.hip): lane = (row = row0 + l%16, chunk = s*4 + l/16)
def kc32(r, c):  # 64-byte rows (BK = 32), kc_swz<32>
    return r * 16 + ((c ^ ((0x1230 >> (4 * ((r >> 2) & 3))) & 3)) << 2)
def kc64(r, c):  # 128-byte rows (BK = 64), lds_kc
    return r * 32 + ((c ^ (r & 7)) << 2)
print("kc32 read (want 4):", [cost([kc32(row0 + l % 16, l // 16) for l in range(64)]) for row0 in (0, 16, 32, 48)])
print("kc64 read (want 4):", [cost([kc64(row0 + l % 16, s * 4 + l // 16) for l in range(64)]) for row0 in (0, 16) for s in (0, 1)])
# search: best 2-bit XOR key function of the row for 64-byte rows
best = None
for keys in itertools.product(range(4), repeat=16):
    if keys[0] != 0: continue
    f = lambda r, c: r * 16 + ((c ^ keys[r % 16]) << 2)
    cst = cost([f(l % 16, l // 16) for l in range(64)])
    if best is None or cst < best[0]:
        best = (cst, keys)
        if cst == 4: break
print("best 64-byte-row key table (period 16):", best)
