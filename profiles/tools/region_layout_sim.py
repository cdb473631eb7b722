"""CPU check of k_decode_region's addressing (host tables of region_geometry + the kernel's index arithmetic)."""
import itertools

def geometry(jx, jy, jz, X=256, Y=256):
    Pmap = [None] * 6
    isx = [False] * 6
    nx = 0
    for k in range(4):
        qb = 3 * k + jx - 2
        if 0 <= qb < 6:
            isx[qb] = True; Pmap[qb] = nx; nx += 1
    assert nx == 2
    nxt = 2
    for qb in range(6):
        if not isx[qb]:
            Pmap[qb] = nxt; nxt += 1
    lanePos = [None] * 6
    for qb in range(6): lanePos[Pmap[qb]] = qb
    parkP = [sum(1 << Pmap[i] for i in range(4) if (gg >> i) & 1) for gg in range(16)]
    parkS = [(1 << Pmap[4] if sv & 1 else 0) | (1 << Pmap[5] if sv & 2 else 0) for sv in range(4)]
    def contrib(qb):
        if qb < 6: return 1 << Pmap[qb]
        i = qb - 6
        return (1 << (6 + i)) | ((1 << (2 + i)) if i < 3 else 0)
    bits = []
    for ax in (1, 2):
        for k in range(4):
            rb = 3 * k + (jy if ax == 1 else jz)
            bits.append(dict(addr=0 if rb < 2 else contrib(rb - 2), byte=(1 << rb) if rb < 2 else 0,
                             out=(1 << k) * X if ax == 1 else (1 << k) * X * Y, ax=ax, k=k))
    used = [False] * 8; order = []
    def take(i): order.append(bits[i]); used[i] = True
    free0 = free1 = top = -1
    for i in range(8):
        if bits[i]['addr'] == 32 and top < 0: top = i
        elif (bits[i]['addr'] & 0x3C) == 0:
            if free0 < 0: free0 = i
            elif free1 < 0: free1 = i
    if free0 >= 0: take(free0)
    else:
        for i in range(8):
            if not used[i] and i != top and i != free1: take(i); break
    if top >= 0: take(top)
    else:
        for i in range(8):
            if not used[i] and i != free1: take(i); break
    if free1 >= 0: take(free1)
    else:
        for i in range(8):
            if not used[i]: take(i); break
    for i in range(8):
        if not used[i]: take(i)
    xr = [0, 0, 0, 0]
    if jx < 2: xr[1] = contrib(3 * 3 + jx - 2)
    else:
        xr[1] = contrib(6); xr[2] = contrib(9); xr[3] = xr[1] ^ xr[2]
    return dict(Pmap=Pmap, lanePos=lanePos, parkP=parkP, parkS=parkS, order=order, xr=xr, jx=jx, jy=jy, jz=jz, X=X, Y=Y, contrib=contrib)

def canon(G, q):
    P = sum(1 << G['Pmap'][i] for i in range(6) if (q >> i) & 1)
    s = q >> 6
    return 64 * s + (P ^ ((s & 7) << 2))

def rank_of(G, x, y, z):
    r = 0
    for k in range(4):
        r |= ((x >> k) & 1) << (3 * k + G['jx'])
        r |= ((y >> k) & 1) << (3 * k + G['jy'])
        r |= ((z >> k) & 1) << (3 * k + G['jz'])
    return r

def check(jx, jy, jz):
    G = geometry(jx, jy, jz)
    # park
    for L in range(64):
        laneTerm = 64 * (L >> 2) + (G['parkS'][L & 3] ^ (((L >> 2) & 7) << 2))
        for gg in range(16):
            assert laneTerm ^ G['parkP'][gg] == canon(G, 16 * L + gg), ('park', L, gg)
    # park bank conflicts: per gg, half-waves
    worst = 0
    for gg in range(16):
        for half in range(2):
            banks = {}
            for L in range(32 * half, 32 * half + 32):
                laneTerm = 64 * (L >> 2) + (G['parkS'][L & 3] ^ (((L >> 2) & 7) << 2))
                b = (laneTerm ^ G['parkP'][gg]) % 32
                banks[b] = banks.get(b, 0) + 1
            worst = max(worst, max(banks.values()))
    # steps
    seen = set()
    for s in range(16):
        for lane in range(64):
            qlow = sum(((lane >> i) & 1) << G['lanePos'][i] for i in range(6))
            addr = 64 * s + (lane ^ ((s & 7) << 2))
            assert addr == canon(G, 64 * s + qlow), ('step', s, lane)
            seen.add(addr)
    assert len(seen) == 1024
    # gather
    written = {}
    slotworst = 0
    for wave in range(8):
        for it in range(4):
            groups = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
            groups += [[l + 32 for l in g] for g in groups]
            lane_addr = {}
            for lane in range(64):
                c = lane & 7
                comb = (lane >> 3) | (it << 3) | (wave << 5)
                addr = 0; bsel = 0; oo = 0
                for i in range(8):
                    if (comb >> i) & 1:
                        addr ^= G['order'][i]['addr']; bsel |= G['order'][i]['byte']; oo += G['order'][i]['out']
                y = (oo // G['X']) % G['Y']; z = oo // (G['X'] * G['Y'])
                assert 0 <= y < 16 and 0 <= z < 16
                assert addr % 4 == 0
                lane_addr[lane] = c * 1028 + addr
                out = []
                if jx < 2:
                    b0 = bsel; b1 = bsel | (1 << jx)
                    for rd in range(2):
                        a0 = addr ^ (G['xr'][1] if rd else 0)
                        words = [a0, a0 + 1, a0 + 2, a0 + 3]
                        # dword: perm(P.y, P.x): bytes [x.b0, x.b1, y.b0, y.b1]
                        out += [(words[0], b0), (words[0], b1), (words[1], b0), (words[1], b1)]
                        out += [(words[2], b0), (words[2], b1), (words[3], b0), (words[3], b1)]
                else:
                    for rd in range(4):
                        a0 = addr ^ G['xr'][rd]
                        out += [(a0, bsel), (a0 + 1, bsel), (a0 + 2, bsel), (a0 + 3, bsel)]
                assert len(out) == 16
                for x in range(16):
                    r = rank_of(G, x, y, z)
                    assert out[x] == (canon(G, r >> 2), r & 3), ('gather', jx, wave, it, lane, x, out[x], canon(G, r >> 2), r & 3)
                key = (c, y, z)
                assert key not in written
                written[key] = 1
            for g in groups:
                slots = {}
                for l in g:
                    sl = (lane_addr[l] // 4) % 16
                    slots.setdefault(sl, set()).add(lane_addr[l])
                slotworst = max(slotworst, max(len(v) for v in slots.values()))
    assert len(written) == 8 * 256
    print('jx,jy,jz', jx, jy, jz, 'ok; park store worst bank multiplicity', worst, '; gather b128 worst distinct addresses per slot', slotworst)

for perm in itertools.permutations(range(3)):
    check(*perm)
