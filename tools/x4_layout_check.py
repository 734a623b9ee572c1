"""Host-side model of the exchange between the two register layouts of the wave-level FHT ("x4": ds_write_addtid_b32
chunks + wide reads, xgpr_amd/csrc/wave_sorf.inc).  Checks (1) that the two exchanges move tile element t between
layout R (lane = t >> 4, register = t & 15) and layout C (lane = t & 63, register = t >> 6), (2) that every LDS access
pattern is conflict-free under the per-instruction lane groups of MI355X_MICROARCH.md (LDS).
    python tools/x4_layout_check.py"""
G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 += [[l + 32 for l in g] for g in G128]
G32 = [list(range(32)), list(range(32, 64))]
X4_DWORDS = 1088
A = lambda j: 66 * j                                             # R -> C: chunk of register j (dwords)
B = lambda r: 64 * (4 * (r & 3) + (r >> 2)) + 4 * (r & 3)         # C -> R: chunk of register r

def conflicts_b128(addr):       # dword addresses (16-byte aligned), 64 banks, four groups of 16 lanes
    worst = 1
    for grp in G128:
        per_bank = {}
        for l in grp:
            a = addr(l)
            assert a % 4 == 0
            for k in range(4):
                per_bank.setdefault((a + k) % 64, set()).add(a + k)
        worst = max(worst, max(len(s) for s in per_bank.values()))
    return worst
def conflicts_b32(addr):        # 32 banks, two groups of 32 lanes
    worst = 1
    for grp in G32:
        per_bank = {}
        for l in grp:
            a = addr(l)
            per_bank.setdefault(a % 32, set()).add(a)
        worst = max(worst, max(len(s) for s in per_bank.values()))
    return worst

# R -> C: register j of lane p = t >> 4 stored at A(j) + p; C lane e = 16 b + j reads r = 0..15 at A(j) + b + 4 r
buf = {}
for t in range(1024):
    a = A(t & 15) + (t >> 4); assert a not in buf and a < X4_DWORDS; buf[a] = t
for e in range(64):
    b, j = e >> 4, e & 15
    for r in range(16):
        t = buf[A(j) + b + 4 * r]; assert (t & 63) == e and (t >> 6) == r
for r in range(16):
    assert conflicts_b32(lambda l: A(l & 15) + (l >> 4) + 4 * r) == 1
# C -> R: register r of lane e = t & 63 stored at B(r) + e; R lane p = b + 4 r reads j = 4 g .. 4 g + 3 at B(r) + 16 b + 4 g
buf = {}
for t in range(1024):
    a = B(t >> 6) + (t & 63); assert a not in buf and a < X4_DWORDS; buf[a] = t
for p in range(64):
    b, r = p & 3, p >> 2
    for j in range(16):
        t = buf[B(r) + 16 * b + j]; assert (t >> 4) == p and (t & 15) == j
for g in range(4):
    assert conflicts_b128(lambda l: B(l >> 2) + 16 * (l & 3) + 4 * g) == 1
# first exchange of a tile with N = P / 64 distinct input registers: R lane p reads chunk (p >> 2) mod N
for N in (2, 4, 8):
    for g in range(4):
        assert conflicts_b128(lambda l: B((l >> 2) & (N - 1)) + 16 * (l & 3) + 4 * g) == 1, N
# images the fused matvec keeps beside it, 64-byte rows read with four ds_read_b128: chi (one row per lane, group g at slot
# g ^ row_swz(lane)) and the datapoint's row (lane l reads row l mod (P / 16), group g at slot g ^ x_swz(row))
row_swz = lambda p: ((p >> 1) & 3) ^ ((p >> 4) & 3)
x_swz = lambda p: (p >> 2) & 3
for g in range(4):
    assert conflicts_b128(lambda l: 16 * l + 4 * (g ^ row_swz(l))) == 1
for P in (128, 256, 512, 1024):
    for g in range(4):
        assert conflicts_b128(lambda l: 16 * (l & (P // 16 - 1)) + 4 * (g ^ x_swz(l & (P // 16 - 1)))) == 1, P
# P <= 256, the transposed-columns layout C2 (lane = 16 t[9:8] + t[3:0], register = t[7:4]): R <-> C2 is the same map both ways --
# register i of all lanes stored at D(i) + lane; lane l reads its sixteen values at D(l & 15) + 16 (l >> 4) + 0 .. 15
D = lambda i: 68 * (4 * (i & 3) + ((0x3102 >> (4 * (i >> 2))) & 15))
for name, lane_of, reg_of, lane_to, reg_to in (
        ("R -> C2", lambda t: t >> 4, lambda t: t & 15, lambda t: 16 * (t >> 8) + (t & 15), lambda t: (t >> 4) & 15),
        ("C2 -> R", lambda t: 16 * (t >> 8) + (t & 15), lambda t: (t >> 4) & 15, lambda t: t >> 4, lambda t: t & 15)):
    buf = {}
    for t in range(1024):
        a = D(reg_of(t)) + lane_of(t); assert a not in buf and a < X4_DWORDS, name; buf[a] = t
    for l in range(64):
        for r in range(16):
            t = buf[D(l & 15) + 16 * (l >> 4) + r]; assert lane_to(t) == l and reg_to(t) == r, name
for g in range(4):
    assert conflicts_b128(lambda l: D(l & 15) + 16 * (l >> 4) + 4 * g) == 1
assert max(D(i) for i in range(16)) + 64 <= X4_DWORDS
# ... and the C2 element index the kernels use: (lane, r) -> 256 (lane >> 4) + 16 r + (lane & 15)
assert sorted(((l >> 4) << 8) | (r << 4) | (l & 15) for l in range(64) for r in range(16)) == list(range(1024))
print("x4 layout: exchanges consistent, all access patterns conflict-free; buffer", X4_DWORDS * 4, "bytes per wave")
