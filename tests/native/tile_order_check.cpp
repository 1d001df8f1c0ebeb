// host check of aurora_amd/csrc/tile_order.h: every (nbm, nbn, G) must map block ids 0 .. nbm * nbn - 1 onto every tile exactly once,
// and inside whole blocks an XCD's tiles of a round must form its 4 x sn sub-block.  g++ -O2 -std=c++17 -I aurora_amd/csrc ...
#include <cstdio>
#include <vector>
#include "tile_order.h"

int main() {
    long checked = 0;
    const int grids[] = {256, 128, 248, 64, 32, 96, 160, 224, 8, 40};
    for (int G : grids)
        for (int nbm = 1; nbm <= 70; nbm += (nbm < 20 ? 1 : 3))
            for (int nbn = 1; nbn <= 90; nbn += (nbn < 20 ? 1 : 7)) {
                TileOrder o;
                tile_order_init(o, nbm, nbn, G);
                std::vector<int> seen(nbm * nbn, 0);
                for (int bid = 0; bid < nbm * nbn; ++bid) {
                    int bm = -1, bn = -1;
                    tile_of_bid(o, bid, bm, bn);
                    if (bm < 0 || bm >= nbm || bn < 0 || bn >= nbn) {
                        printf("FAIL out of range G=%d nbm=%d nbn=%d bid=%d -> (%d, %d)\n", G, nbm, nbn, bid, bm, bn);
                        return 1;
                    }
                    if (seen[bm * nbn + bn]++) {
                        printf("FAIL duplicate G=%d nbm=%d nbn=%d bid=%d -> (%d, %d)\n", G, nbm, nbn, bid, bm, bn);
                        return 1;
                    }
                    if (bid < o.full) {          // the sub-block of XCD bid % 8 in round bid / G
                        const int blk = bid / G, bi = blk / o.nfn, bj = blk % o.nfn, xcd = bid & 7;
                        const int m0 = bi * o.bmt + (xcd / o.xn) * 4, n0 = bj * o.bnt + (xcd % o.xn) * o.sn;
                        if (bm < m0 || bm >= m0 + 4 || bn < n0 || bn >= n0 + o.sn) {
                            printf("FAIL sub-block G=%d nbm=%d nbn=%d bid=%d\n", G, nbm, nbn, bid);
                            return 1;
                        }
                    }
                }
                ++checked;
            }
    printf("ok %ld shapes\n", checked);
    return 0;
}
