"""LDS bank-conflict model of the matrix-core passes of nsk3_mfma_ops.hpp and the search behind PadLay<8>: for every tile stage
(one writer pattern, one reader pattern) the stride triple (slowest, middle, fastest index) with the fewest LDS cycles.
Bank rules of the CDNA4 guide: ds_read_b64 is served in two 32-lane groups over 64 banks (slot = double index mod 32), ds_write_b64
in four 16-lane groups over 32 banks (slot = double index mod 16); each extra distinct address on a busy slot of a group costs a cycle.
    python scripts/lds_bank_search.py [lx1=8] [extent limit of the N^3 tile=640]"""
import sys
def cost(addrs_by_lane, kind):
    if kind == "r": groups = [range(0,32), range(32,64)]; mod = 32
    else: groups = [range(g*16, g*16+16) for g in range(4)]; mod = 16
    c = 0
    for g in groups:
        slots = {}
        for l in g:
            a = addrs_by_lane[l]
            if a is None: continue
            slots.setdefault(a % mod, set()).add(a)
        c += max([len(s) for s in slots.values()] + [0 if not slots else 1])
    return c
def rd(K, cols, ks):
    """read cost of a pass: cols = list of column base addresses (len NCOL), contracted stride ks"""
    NCOL = len(cols); KQ = (K + 3)//4; c = 0
    for tile in range((NCOL + 15)//16):
        for q in range(KQ):
            a = []
            for lane in range(64):
                n = tile*16 + (lane & 15); k = 4*q + (lane >> 4)
                a.append(cols[n] + k*ks if (n < NCOL and k < K) else None)
            c += cost(a, "r")
    return c
def wr(MR, cols, ms):
    NCOL = len(cols); c = 0
    for tile in range((NCOL + 15)//16):
        for r in range((MR + 3)//4):
            a = []
            for lane in range(64):
                n = tile*16 + (lane & 15); m = (lane >> 4) + 4*r
                a.append(cols[n] + m*ms if (n < NCOL and m < MR) else None)
            c += cost(a, "w")
    return c
def lin_w(addr):      # thread-per-node write, addr = list over tid
    c = 0
    for w in range((len(addr) + 63)//64):
        a = [addr[w*64 + l] if w*64 + l < len(addr) else None for l in range(64)]
        c += cost(a, "w")
    return c
def lin_r(addr):
    c = 0
    for w in range((len(addr) + 63)//64):
        a = [addr[w*64 + l] if w*64 + l < len(addr) else None for l in range(64)]
        c += cost(a, "r")
    return c
def cols_for(kind, dims, s):
    D0, D1, D2 = dims; s0, s1, s2 = s
    if kind == "t": return [p*s1 + q*s2 for p in range(D1) for q in range(D2)], s0, D0
    if kind == "s": return [p*s0 + q*s2 for p in range(D0) for q in range(D2)], s1, D1
    if kind == "r": return [p*s0 + q*s1 for p in range(D0) for q in range(D1)], s2, D2
def lin_addr(dims, s, sub=None):
    D0, D1, D2 = dims
    if sub is None: return [a*s[0] + b*s[1] + c*s[2] for a in range(D0) for b in range(D1) for c in range(D2)]
    o, (E0, E1, E2) = sub
    return [(a+o)*s[0] + (b+o)*s[1] + (c+o)*s[2] for a in range(E0) for b in range(E1) for c in range(E2)]
def injective(dims, s, lim):
    seen = set()
    for a in range(dims[0]):
        for b in range(dims[1]):
            for c in range(dims[2]):
                x = a*s[0] + b*s[1] + c*s[2]
                if x in seen or x >= lim: return False
                seen.add(x)
    return True
def search(name, dims, wkind, rkind, lim, s2max=3, sub=None):
    best = None
    for s2 in range(1, s2max + 1):
        for s1 in range(1, lim//max(1, dims[1]-1) + 1):
            for s0 in range(1, lim//max(1, dims[0]-1) + 1):
                s = (s0, s1, s2)
                ext = (dims[0]-1)*s0 + (dims[1]-1)*s1 + (dims[2]-1)*s2 + 1
                if ext > lim or not injective(dims, s, lim): continue
                c = 0
                if wkind == "lin": c += lin_w(lin_addr(dims, s))
                else:
                    cols, ms, MR = cols_for(wkind, dims, s); c += wr(MR, cols, ms)
                if rkind == "lin": c += lin_r(lin_addr(dims, s, sub))
                else:
                    cols, ks, K = cols_for(rkind, dims, s); c += rd(K, cols, ks)
                key = (c, ext)
                if best is None or key < best[0]: best = (key, s)
    cur = (dims[1]*dims[2], dims[2], 1)
    c0 = 0
    if wkind == "lin": c0 += lin_w(lin_addr(dims, cur))
    else:
        cols, ms, MR = cols_for(wkind, dims, cur); c0 += wr(MR, cols, ms)
    if rkind == "lin": c0 += lin_r(lin_addr(dims, cur, sub))
    else:
        cols, ks, K = cols_for(rkind, dims, cur); c0 += rd(K, cols, ks)
    print("%-28s dims %s write %-3s read %-3s: now %4d -> best %4d with strides %s extent %d" % (name, dims, wkind, rkind, c0, best[0][0], best[1], best[0][1]))
    return best[1]
if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    M = N - 2
    T = (N, N, N)
    lim = int(sys.argv[2]) if len(sys.argv) > 2 else 640
    print("FDM tile")
    search("L0 fill -> r", T, "lin", "r", lim)
    search("L1 r -> s", T, "r", "s", lim)
    search("L2 s -> t", T, "s", "t", lim)
    search("L3 t -> t", T, "t", "t", lim)
    search("L4 t -> s", T, "t", "s", lim)
    search("L5 s -> r", T, "s", "r", lim)
    search("L6 r -> restriction", T, "r", "lin", lim, sub=(1, (M, M, M)))
    print("D^T")
    search("sP products -> t", (M, M, M), "lin", "t", 256)
    search("sC t -> s", (N, M, M), "t", "s", 340)
    search("sE s -> r", (N, N, M), "s", "r", 470)
    search("stage r -> lin", T, "r", "lin", 3*340)
