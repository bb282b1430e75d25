"""Lane-level numpy emulation of csrc/tn_taps.hip (index arithmetic only: LDS image, swizzle, transposed reads, ring,
edge masks, MFMA operand layout).  Runs on the CPU; used to check the kernel's addressing before it meets a GPU.

    python tools/emulate_tn_taps.py
"""
import numpy as np


def taps_hash(row):
    return ((row >> 1) & 1) | (((row >> 3) & 1) << 1)


def layout(N, H, W, dil):
    Wp = W + dil
    IP = (H * Wp + 7) // 8 * 8
    ln = (N * IP + 63) // 64 * 64
    return Wp, IP, ln


def build_table(N, H, W, Wp, IP, ln):
    tab = np.full(ln, -1, np.int64)
    for q in range(ln):
        n, rem = divmod(q, IP)
        y, x = divmod(rem, Wp)
        if n < N and y < H and x < W:
            tab[q] = (n * H + y) * W + x
    return tab


def run_workgroup(args, blockIdx, gridDim, C, colsum, RL=2, HALO=1):
    (A, B, tab, NA, Cg, lda, ldg, ldc, Wp, IP8, top_lo, bot_hi, dil, nchunks, cps) = args
    ROWB, CHB = 128, 64 * 128
    XRING = (1 << RL) * CHB
    XMASK = XRING - 1
    ARING = 4 * CHB
    sX, sA, sM = 0, XRING, XRING + ARING
    lds = np.zeros((XRING + ARING) // 2, np.float64)  # one slot per bf16 element
    lds[:] = np.nan                                    # garbage until written
    masks = {}

    tiles_b, tiles_a = Cg >> 6, (NA + 63) >> 6
    total = gridDim
    xq, xr, xcd = total >> 3, total & 7, blockIdx & 7
    vb = xcd * xq + min(xcd, xr) + (blockIdx >> 3)
    ntiles = tiles_a * tiles_b
    split, tile = divmod(vb, ntiles)
    tile_a, tile_b = divmod(tile, tiles_b)
    na0, cb0 = tile_a * 64, tile_b * 64
    c_begin = split * cps
    c_end = min(nchunks, c_begin + cps)
    if c_begin >= c_end:
        return vb

    lane = np.arange(64)
    srow, pslot, half = lane >> 3, (lane & 7) >> 1, lane & 1
    l15, lg = lane & 15, lane >> 4
    frow = lg * 8 + (l15 >> 2)

    def stage(cc, with_a):
        for wave in range(4):
            for jj in range(2):
                srowc = (wave * 2 + jj) * 8 + srow
                scol = (((pslot ^ taps_hash(srowc)) << 1) + half) * 8
                ent = tab[cc * 64 + srowc] if 0 <= cc < nchunks else np.full(64, -1)
                dX = sX + (cc & ((1 << RL) - 1)) * CHB + wave * 2048 + jj * 1024
                dA = sA + (cc & 3) * CHB + wave * 2048 + jj * 1024
                for l in range(64):
                    ok = ent[l] >= 0
                    dst = (dX + l * 16) // 2
                    if ok:
                        off = ent[l] * ldg + cb0 + scol[l]
                        lds[dst:dst + 8] = B[off:off + 8]
                    else:
                        lds[dst:dst + 8] = 0
                    if with_a:
                        dst = (dA + l * 16) // 2
                        if ok and na0 + scol[l] < NA:
                            off = ent[l] * lda + na0 + scol[l]
                            lds[dst:dst + 8] = A[off:off + 8]
                        else:
                            lds[dst:dst + 8] = 0

    def tr_read(addr):
        """addr[64] byte addresses -> [64, 4] values (ds_read_b64_tr_b16)."""
        out = np.zeros((64, 4))
        for l in range(64):
            g, i = l >> 4, l & 15
            for j in range(4):
                src = g * 16 + j * 4 + (i >> 2)
                a = addr[src] + (i & 3) * 2
                assert a % 2 == 0 and addr[src] % 8 == 0
                out[l, j] = lds[a // 2]
        return out

    def tr_read2(p0, p1):
        return np.concatenate([tr_read(p0), tr_read(p1)], axis=1)  # [64, 8]

    def mfma(fa, fb, acc):
        # A[row=l15][k=lg*8+e], B[k][col=l15]; acc[lane=(col, lg)][q] = D[lg*4+q][col]
        Am = np.zeros((16, 32))
        Bm = np.zeros((32, 16))
        for l in range(64):
            Am[l & 15, (l >> 4) * 8:(l >> 4) * 8 + 8] = fa[l]
            Bm[(l >> 4) * 8:(l >> 4) * 8 + 8, l & 15] = fb[l]
        D = Am @ Bm
        for l in range(64):
            for q in range(4):
                acc[l, q] += D[(l >> 4) * 4 + q, l & 15]

    # mask table
    for e in range(IP8):
        mt = np.zeros(8)
        mb = np.zeros(8)
        for j in range(8):
            q = e * 8 + j
            mt[j] = 1.0 if q >= top_lo else 0.0
            mb[j] = 1.0 if q < bot_hi else 0.0
        masks[e] = (mt, mb)

    for cc in range(-HALO, HALO + 1):
        stage(c_begin + cc, cc >= 0 and c_begin + cc < c_end)

    acc = np.zeros((4, 4, 9, 64, 4))  # wave, i, tap, lane, q
    accs = np.zeros((4, 64, 4))
    offA = [frow * ROWB + ((i ^ taps_hash(frow)) << 5) + (l15 & 3) * 8 for i in range(4)]
    ring_rows = 64 << RL
    ment = (c_begin * 8 + lg) % IP8
    ment_step = 4 % IP8
    do_colsum = colsum is not None and tile_b == 0
    for c in range(c_begin, c_end):
        stage(c + HALO + 1, c + HALO + 1 < c_end)
        # NOTE: the kernel issues this DMA before the compute of step c; slot (c+HALO+1) must not be read in step c
        bA = sA + (c & 3) * CHB
        for kk in range(2):
            kofs = ((c * 2 + kk) & ((2 << RL) - 1)) << 12
            mt = np.stack([masks[int(e)][0] for e in ment])
            mb = np.stack([masks[int(e)][1] for e in ment])
            ment = ment + ment_step
            ment = np.where(ment >= IP8, ment - IP8, ment)
            for wave in range(4):
                fa = [tr_read2(bA + offA[i] + kk * 32 * ROWB, bA + offA[i] + kk * 32 * ROWB + 4 * ROWB) for i in range(4)]
                if do_colsum and wave == 0:
                    for i in range(4):
                        mfma(fa[i], np.ones((64, 8)), accs[i])
                for t in range(9):
                    shift = ((t // 3) - 1) * dil * Wp + ((t % 3) - 1) * dil
                    o0 = []
                    for hh in range(2):
                        r0 = (frow + hh * 4 + shift) & (ring_rows - 1)
                        o0.append(r0 * ROWB + ((wave ^ taps_hash(r0)) << 5) + (l15 & 3) * 8)
                    fb = tr_read2(sX + ((o0[0] + kofs) & XMASK), sX + ((o0[1] + kofs) & XMASK))
                    assert not np.isnan(fb).any(), "read of an unwritten LDS element"
                    if t < 3:
                        fb = fb * mt
                    if t >= 6:
                        fb = fb * mb
                    for i in range(4):
                        mfma(fa[i], fb, acc[wave, i, t])

    for wave in range(4):
        for t in range(9):
            for l in range(64):
                col = t * Cg + cb0 + wave * 16 + (l & 15)
                for i in range(4):
                    for q in range(4):
                        row = na0 + i * 16 + (l >> 4) * 4 + q
                        if row < NA:
                            C[row, col] += acc[wave, i, t, l, q]
    if do_colsum:
        for l in range(0, 64, 16):
            for i in range(4):
                for q in range(4):
                    row = na0 + i * 16 + (l >> 4) * 4 + q
                    if row < NA:
                        colsum[row] += accs[i, l, q]
    return vb


def reference(dy, x, N, H, W, Cin, Cout, dil):
    dyv = dy.reshape(N, H, W, Cout)
    xv = x.reshape(N, H, W, Cin)
    dw = np.zeros((Cout, 3, 3, Cin))
    for r in range(3):
        for s in range(3):
            for h in range(H):
                for w in range(W):
                    hi, wi = h + (r - 1) * dil, w + (s - 1) * dil
                    if 0 <= hi < H and 0 <= wi < W:
                        dw[:, r, s, :] += dyv[:, h, w, :].T @ xv[:, hi, wi, :]
    return dw.reshape(Cout, 9 * Cin), dyv.reshape(-1, Cout).sum(0)


def check(N, H, W, Cin, Cout, dil, splits, seed=0):
    rng = np.random.default_rng(seed)
    P = N * H * W
    lda, ldg = Cout, Cin
    dy = rng.integers(-3, 4, (P, lda)).astype(np.float64)
    x = rng.integers(-3, 4, (P, ldg)).astype(np.float64)
    Wp, IP, ln = layout(N, H, W, dil)
    assert dil * Wp + dil <= 64 and IP >= 32
    tab = build_table(N, H, W, Wp, IP, ln)
    nchunks = ln // 64
    cps = -(-nchunks // splits)
    splits = -(-nchunks // cps)
    tiles = -(-Cout // 64) * (Cin // 64)
    C = np.zeros((Cout, 9 * Cin))
    cs = np.zeros(Cout)
    args = (dy.ravel(), x.ravel(), tab, Cout, Cin, lda, ldg, 9 * Cin, Wp, IP // 8, dil * Wp, (H - dil) * Wp, dil,
            nchunks, cps)
    seen = set()
    for b in range(tiles * splits):
        seen.add(run_workgroup(args, b, tiles * splits, C, cs))
    assert seen == set(range(tiles * splits)), "work map is not a permutation"
    ref, refcs = reference(dy, x, N, H, W, Cin, Cout, dil)
    err = np.abs(C - ref).max()
    errc = np.abs(cs - refcs).max()
    print("N=%d H=%d W=%d Cin=%d Cout=%d dil=%d splits=%d  max|dW err|=%g  max|db err|=%g" %
          (N, H, W, Cin, Cout, dil, splits, err, errc))
    assert err == 0 and errc == 0


if __name__ == "__main__":
    check(3, 4, 9, 64, 64, 1, 1)
    check(5, 4, 33, 64, 64, 1, 3)
    check(2, 8, 32, 128, 72, 1, 2)
    check(3, 6, 10, 64, 64, 2, 2)
    print("ok")
