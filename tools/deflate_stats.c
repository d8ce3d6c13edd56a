/* deflate_stats -- DEVELOPER TOOL: statistics of the DEFLATE streams inside the BGZF members of a file (what thj_k_inflate has to
 * decode): deflate blocks per member, symbols per member, literal / match mix, match length and distance histograms.  Own
 * bit-by-bit canonical decoder (RFC 1951), no zlib.   gcc -O2 -o tools/bin/deflate_stats tools/deflate_stats.c */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

typedef struct { const uint8_t* p; size_t n; size_t bit; } Bits;
static uint32_t getb(Bits* b, int n) { uint32_t v = 0; for (int i = 0; i < n; ++i, ++b->bit) v |= (uint32_t)((b->p[b->bit >> 3] >> (b->bit & 7)) & 1) << i; return v; }
typedef struct { uint16_t cnt[16], sym[320]; } Huff;
static void build(Huff* h, const uint8_t* lens, int n) {
    uint16_t offs[16]; memset(h->cnt, 0, sizeof h->cnt);
    for (int i = 0; i < n; ++i) h->cnt[lens[i]]++;
    offs[1] = 0; for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + h->cnt[l];
    for (int i = 0; i < n; ++i) if (lens[i]) h->sym[offs[lens[i]]++] = (uint16_t)i;
}
static int dec(Bits* b, const Huff* h, int* used) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; ++l) {
        code |= (int)getb(b, 1);
        int c = h->cnt[l];
        if (code - c < first) { *used = l; return h->sym[index + (code - first)]; }
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}

static uint64_t n_members, n_blocks, n_lit, n_match, n_matchbytes, n_out, n_comp, hdr_bits, blk_type[3];
static uint64_t dist_hist[16], len_hist[10], litlen_codelen[16], dist_codelen[16];
static uint64_t run_lit_hist[8];     /* literal run lengths between matches: 0,1,2,3-4,5-8,9-16,17-32,>32 */
static uint64_t maxlitlen_hist[16], maxdistlen_hist[16];

static int bucket(uint32_t d) { int k = 0; while ((1u << (k + 1)) <= d && k < 15) ++k; return k; }

static int inflate_member(const uint8_t* p, size_t n, uint32_t isize) {
    Bits b = {p, n, 0};
    uint32_t outp = 0; int last = 0;
    while (!last) {
        last = (int)getb(&b, 1); int type = (int)getb(&b, 2);
        ++n_blocks; if (type < 3) ++blk_type[type];
        if (type == 0) { b.bit = (b.bit + 7) & ~7ull; uint32_t len = getb(&b, 16); getb(&b, 16); b.bit += 8ull * len; outp += len; n_lit += len; continue; }
        if (type == 3) return -1;
        Huff lh, dh; uint8_t lens[320];
        size_t h0 = b.bit;
        if (type == 1) {
            for (int i = 0; i < 144; ++i) lens[i] = 8; for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7; for (int i = 280; i < 288; ++i) lens[i] = 8;
            build(&lh, lens, 288); for (int i = 0; i < 30; ++i) lens[i] = 5; build(&dh, lens, 30);
        } else {
            int hlit = (int)getb(&b, 5) + 257, hdist = (int)getb(&b, 5) + 1, hclen = (int)getb(&b, 4) + 4;
            uint8_t cl[19] = {0}; for (int i = 0; i < hclen; ++i) cl[CLORD[i]] = (uint8_t)getb(&b, 3);
            Huff ch; build(&ch, cl, 19);
            int i = 0, u;
            while (i < hlit + hdist) {
                int s = dec(&b, &ch, &u); if (s < 0) return -1;
                if (s < 16) lens[i++] = (uint8_t)s;
                else { int rep, val = 0; if (s == 16) { val = lens[i - 1]; rep = 3 + (int)getb(&b, 2); } else if (s == 17) rep = 3 + (int)getb(&b, 3); else rep = 11 + (int)getb(&b, 7);
                    while (rep--) lens[i++] = (uint8_t)val; }
            }
            build(&lh, lens, hlit); build(&dh, lens + hlit, hdist);
            int ml = 0, md = 0; for (int k = 0; k < hlit; ++k) if (lens[k] > ml) ml = lens[k]; for (int k = 0; k < hdist; ++k) if (lens[hlit + k] > md) md = lens[hlit + k];
            maxlitlen_hist[ml]++; maxdistlen_hist[md]++;
        }
        hdr_bits += b.bit - h0;
        uint32_t run = 0;
        for (;;) {
            int u; int s = dec(&b, &lh, &u); if (s < 0) return -1;
            litlen_codelen[u]++;
            if (s < 256) { ++outp; ++n_lit; ++run; continue; }
            if (s == 256) break;
            s -= 257; uint32_t len = LBASE[s] + getb(&b, LEXT[s]);
            int ds = dec(&b, &dh, &u); if (ds < 0) return -1;
            dist_codelen[u]++;
            uint32_t dist = DBASE[ds] + getb(&b, DEXT[ds]);
            ++n_match; n_matchbytes += len; outp += len;
            dist_hist[bucket(dist)]++;
            len_hist[len <= 3 ? 0 : len <= 4 ? 1 : len <= 6 ? 2 : len <= 8 ? 3 : len <= 12 ? 4 : len <= 16 ? 5 : len <= 32 ? 6 : len <= 64 ? 7 : len <= 128 ? 8 : 9]++;
            run_lit_hist[run == 0 ? 0 : run == 1 ? 1 : run == 2 ? 2 : run <= 4 ? 3 : run <= 8 ? 4 : run <= 16 ? 5 : run <= 32 ? 6 : 7]++;
            run = 0;
        }
    }
    if (outp != isize) return -2;
    n_out += outp;
    return 0;
}

int main(int argc, char** argv) {
    for (int a = 1; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb"); if (!f) { perror(argv[a]); return 1; }
        fseek(f, 0, SEEK_END); size_t n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
        uint8_t* d = malloc(n + 8); if (fread(d, 1, n, f) != n) return 1; fclose(f); memset(d + n, 0, 8);
        size_t off = 0;
        while (off + 18 < n) {
            uint32_t bsize = (uint32_t)(d[off + 16] | d[off + 17] << 8) + 1;
            uint32_t isize; memcpy(&isize, d + off + bsize - 4, 4);
            int rc = inflate_member(d + off + 18, bsize - 26, isize);
            if (rc) { fprintf(stderr, "%s: member at %zu: rc %d\n", argv[a], off, rc); return 1; }
            ++n_members; n_comp += bsize; off += bsize;
        }
        free(d);
    }
    uint64_t sy = n_lit + n_match;
    printf("members %llu  deflate blocks %llu (%.2f per member; stored %llu fixed %llu dynamic %llu)  header bits per dynamic block %.0f\n", (unsigned long long)n_members,
           (unsigned long long)n_blocks, (double)n_blocks / n_members, (unsigned long long)blk_type[0], (unsigned long long)blk_type[1], (unsigned long long)blk_type[2], blk_type[2] ? (double)hdr_bits / blk_type[2] : 0.0);
    printf("inflated %.1f MB, compressed %.1f MB (ratio %.3f); symbols %llu = %.1f per member, %.2f bytes per symbol; literals %.1f %% of symbols, %.1f %% of bytes; mean match %.1f bytes\n",
           n_out / 1e6, n_comp / 1e6, (double)n_comp / n_out, (unsigned long long)sy, (double)sy / n_members, (double)n_out / sy, 100.0 * n_lit / sy, 100.0 * n_lit / n_out, (double)n_matchbytes / n_match);
    printf("match distance (share of matches, cumulative):");
    { uint64_t c = 0; for (int k = 0; k < 16; ++k) { c += dist_hist[k]; printf(" <%u:%.1f%%", 2u << k, 100.0 * c / n_match); } printf("\n"); }
    { static const char* nm[10] = {"3", "4", "5-6", "7-8", "9-12", "13-16", "17-32", "33-64", "65-128", ">128"}; printf("match length:"); for (int k = 0; k < 10; ++k) printf(" %s:%.1f%%", nm[k], 100.0 * len_hist[k] / n_match); printf("\n"); }
    { static const char* nm[8] = {"0", "1", "2", "3-4", "5-8", "9-16", "17-32", ">32"}; printf("literal run before a match:"); for (int k = 0; k < 8; ++k) printf(" %s:%.1f%%", nm[k], 100.0 * run_lit_hist[k] / n_match); printf("\n"); }
    printf("lit/len code length used:"); for (int k = 1; k < 16; ++k) printf(" %d:%.1f%%", k, 100.0 * litlen_codelen[k] / sy); printf("\n");
    printf("dist code length used:"); for (int k = 1; k < 16; ++k) printf(" %d:%.1f%%", k, 100.0 * dist_codelen[k] / n_match); printf("\n");
    printf("max lit/len code length per block:"); for (int k = 1; k < 16; ++k) if (maxlitlen_hist[k]) printf(" %d:%llu", k, (unsigned long long)maxlitlen_hist[k]); printf("\n");
    printf("max dist code length per block:"); for (int k = 1; k < 16; ++k) if (maxdistlen_hist[k]) printf(" %d:%llu", k, (unsigned long long)maxdistlen_hist[k]); printf("\n");
    return 0;
}
