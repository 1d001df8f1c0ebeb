// Order in which the persistent 256x256 GEMM walks its output tiles (gemm256.hip) - plain C++ so that the host-side test
// (tests/test_tile_order.py, g++) can check that it is a bijection.
//
// A grid of G workgroups (G % 32 == 0; workgroup b sits on XCD b % 8) works on G tiles at a time ("round").  A round is ONE compact
// block of the tile space, (xm * 4) x (xn * sn) tiles, split into one 4 x sn sub-block per XCD:
//   * inside an XCD the sub-block's 4 + sn operand tiles are shared through its L2 by the 4 * sn concurrent workgroups (as before);
//   * across XCDs the block's operands are shared in time: the 8 XCDs of a round read the same xm * 4 activation and xn * sn weight
//     tiles within one tile's duration, so all but the first reader can be served by the memory-side cache instead of HBM
//     (256 workgroups: 16 x 16 tiles, 64 MiB of unique operands per 256 tiles instead of 8 x 24 MiB).
// Consecutive rounds walk along N inside a block row (the activation rows stay).  Tiles outside the whole blocks (right and bottom
// strips) follow in super-rows of 4 M-tiles, column-major.  Grids that are not a multiple of 32 use the strip order for everything.
#pragma once
#ifdef __HIPCC__
#define AUR_HD __host__ __device__ __forceinline__
#else
#define AUR_HD inline
#endif

// (Round 3 also had a "hand-down" order - an XCD keeps its band of 4 M-tiles, weight column groups pass from XCD to XCD round by
// round - built on a wrong reading of the memory-side cache; measured neutral in the lab and 0.6 % slower in the bench: removed.)
struct TileOrder {
    int nbm, nbn, G;
    int sn, xn, bmt, bnt;            // sub-block 4 x sn per XCD, XCDs arranged (8 / xn) x xn, block bmt x bnt tiles (bmt * bnt == G)
    int nfm, nfn, full;              // whole blocks along M / N, tiles inside them
    int right;                       // tiles of the right strip: rows [0, nfm * bmt) x cols [nfn * bnt, nbn)
};

AUR_HD void tile_order_init(TileOrder& o, int nbm, int nbn, int G) {
    o.nbm = nbm; o.nbn = nbn; o.G = G;
    o.sn = 0; o.xn = 1; o.bmt = 0; o.bnt = 0; o.nfm = 0; o.nfn = 0; o.full = 0; o.right = 0;
    if (G >= 32 && (G & 31) == 0) {
        o.sn = G >> 5;                                   // (G / 8 tiles per XCD) / 4 rows
        o.xn = o.sn >= 8 ? 2 : 4;
        o.bmt = (8 / o.xn) * 4;
        o.bnt = o.xn * o.sn;
        o.nfm = nbm / o.bmt;
        o.nfn = nbn / o.bnt;
        o.full = o.nfm * o.nfn * G;
        o.right = o.nfm * o.bmt * (nbn - o.nfn * o.bnt);
    }
}

// super-rows of 4 M-tiles, column-major inside, over the rectangle rows [m0, m1) x cols [n0, n1)
AUR_HD void tile_strip(int lid, int m0, int m1, int n0, int n1, int& bm, int& bn) {
    const int wdt = n1 - n0;
    const int sr = lid / (4 * wdt), rem = lid - sr * 4 * wdt;
    const int left = (m1 - m0) - 4 * sr;
    const int rows = left < 4 ? left : 4;
    bn = n0 + rem / rows;
    bm = m0 + 4 * sr + rem % rows;
}

AUR_HD void tile_of_bid(const TileOrder& o, int bid, int& bm, int& bn) {
    if (bid < o.full) {
        const int blk = bid / o.G, j = bid - blk * o.G;
        const int bi = blk / o.nfn, bj = blk - bi * o.nfn;
        const int xcd = j & 7, k = j >> 3;
        const int sx = xcd / o.xn, sy = xcd - sx * o.xn;
        bm = bi * o.bmt + sx * 4 + (k & 3);
        bn = bj * o.bnt + sy * o.sn + (k >> 2);
        return;
    }
    int e = bid - o.full;
    if (e < o.right) {
        tile_strip(e, 0, o.nfm * o.bmt, o.nfn * o.bnt, o.nbn, bm, bn);
        return;
    }
    e -= o.right;
    tile_strip(e, o.nfm * o.bmt, o.nbm, 0, o.nbn, bm, bn);
}
