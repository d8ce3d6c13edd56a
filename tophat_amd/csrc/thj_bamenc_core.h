// thj_bamenc_core.h -- one output alignment as the BAM record print_bamhit writes for it (bwt_map.cpp:1888-2093; GBamRecord's
// constructor and add_aux, common.cpp:1005-1173; the mate fields "*", 0, 0 and MAPQ 255 as there), built from the device record
// (thj_aln) and the read's own BAM record, whose name, packed bases and qualities are copied -- reversed and complemented
// nibble-wise for an antisense alignment (reverse_complement, reads.cpp:189-207: anything but A C G T becomes N).
// Byte for byte what the host encoder (long_spanning_reads_main.cpp, encode_plain_from_raw) writes; the records it cannot
// take -- fusion alignments (two records with XF:Z), MD strings the device record does not hold, a read whose length differs
// from the alignment's -- are reported, and the caller leaves the whole batch to the host encoder.
// Plain per-record functions: the HIP kernels (thj_bamout.hip) call them per thread, tests/hostsim calls them on the CPU.
#pragma once
#include <stdint.h>
#include <string.h>
#include "../../include/thj.h"

#ifndef THJ_DFN
#define THJ_DFN inline
#endif

namespace bamenc {

THJ_DFN uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
THJ_DFN void wr32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }

// cigar op of the device record -> BAM op (set_cigar, common.cpp:1044-1053, upper-cases the lower-case ops)
THJ_DFN uint32_t bam_op(uint32_t op) { return (uint32_t)((0x6543300002211000ull >> (4 * op)) & 15u); }
// bytes of an integer tag's value: the smallest type that holds it (add_aux, common.cpp:1092-1173)
THJ_DFN uint32_t int_bytes(int x) { return x < 0 ? (x >= -127 ? 1u : x >= -32767 ? 2u : 4u) : (x <= 255 ? 1u : x <= 65535 ? 2u : 4u); }
THJ_DFN uint8_t* put_int(uint8_t* o, char t0, char t1, int x) {
    o[0] = (uint8_t)t0; o[1] = (uint8_t)t1;
    if (x < 0) {
        if (x >= -127) { o[2] = 'c'; o[3] = (uint8_t)(int8_t)x; return o + 4; }
        if (x >= -32767) { o[2] = 's'; const uint16_t v = (uint16_t)(int16_t)x; o[3] = (uint8_t)v; o[4] = (uint8_t)(v >> 8); return o + 5; }
        o[2] = 'i';
    } else {
        if (x <= 255) { o[2] = 'C'; o[3] = (uint8_t)x; return o + 4; }
        if (x <= 65535) { o[2] = 'S'; o[3] = (uint8_t)x; o[4] = (uint8_t)(x >> 8); return o + 5; }
        o[2] = 'I';
    }
    const uint32_t v = (uint32_t)x;
    o[3] = (uint8_t)v; o[4] = (uint8_t)(v >> 8); o[5] = (uint8_t)(v >> 16); o[6] = (uint8_t)(v >> 24);
    return o + 7;
}
// bam_reg2bin (bam.h)
THJ_DFN uint32_t reg2bin(int32_t beg, int32_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct Shape { uint32_t size; int32_t rlen, indel; bool spliced, host_only; int64_t rid; };

// size of the record (block_size field included), the read id (atol of the name), and whether the host encoder must take it.
// raw: the read's BAM record after its block_size field.
THJ_DFN Shape record_shape(const thj_aln& a, const uint8_t* raw) {
    Shape s; s.rlen = 0; s.indel = 0; s.spliced = false; s.host_only = false; s.rid = 0;
    for (int k = 0; k < a.n_cigar && k < 16; ++k) {
        const uint32_t op = a.cigar[k] >> 28, len = a.cigar[k] & 0x0FFFFFFFu;
        if (op == 1 || op == 2 || op == 3 || op == 4 || op == 13) s.rlen += (int32_t)len;
        if (op >= 3 && op <= 6) s.indel += (int32_t)len;
        if (op == 11 || op == 12) s.spliced = true;
        if (op >= THJ_CIG_FUSION_FF && op <= THJ_CIG_FUSION_RR) s.host_only = true;
    }
    const uint32_t l_rn = rd32(raw + 8) & 0xFFu, lseq = rd32(raw + 16);
    if (a.n_cigar > 16 || a.md_len == THJ_MD_ON_HOST || (int32_t)lseq != s.rlen || l_rn == 0) s.host_only = true;
    s.size = 36u + l_rn + 4u * a.n_cigar + ((lseq + 1) >> 1) + lseq + (3u + int_bytes((int)a.AS)) + (3u + int_bytes((int)a.XM)) + (3u + int_bytes((int)a.XO)) +
             (3u + int_bytes((int)a.XG)) + (3u + (s.host_only ? 0u : (uint32_t)a.md_len) + 1u) + (3u + int_bytes((int)a.mismatches + s.indel)) + (s.spliced ? 4u : 0u);
    {   // atol(qname)
        const char* q = (const char*)raw + 32;
        bool neg = false; uint32_t k = 0; int64_t v = 0;
        while (k + 1 < l_rn && (q[k] == ' ' || q[k] == '\t')) ++k;
        if (k + 1 < l_rn && q[k] == '-') { neg = true; ++k; } else if (k + 1 < l_rn && q[k] == '+') ++k;
        for (; k + 1 < l_rn && q[k] >= '0' && q[k] <= '9'; ++k) v = v * 10 + (q[k] - '0');
        s.rid = neg ? -v : v;
    }
    return s;
}

// the record's bytes at o (s = record_shape(a, raw), not host_only); tid = the contig's index in the output header
THJ_DFN void record_write(const thj_aln& a, const uint8_t* raw, const Shape& s, int32_t tid, uint8_t* o) {
    const uint32_t l_rn = rd32(raw + 8) & 0xFFu, n_cig_in = rd32(raw + 12) & 0xFFFFu, lseq = rd32(raw + 16);
    const uint8_t* sq = raw + 32 + l_rn + 4 * n_cig_in;
    const uint8_t* ql = sq + ((lseq + 1) >> 1);
    const bool anti = (a.flags & THJ_HIT_ANTISENSE) != 0;
    const int32_t pos = a.left + 1 <= 0 ? -1 : a.left;
    int32_t end = pos;
    for (int i = 0; i < a.n_cigar; ++i) { const uint32_t c = a.cigar[i], op = bam_op(c >> 28); if (op == 0 || op == 2 || op == 3) end += (int32_t)(c & 0x0FFFFFFFu); }
    const uint32_t bin = reg2bin(pos, a.n_cigar == 0 ? pos + 1 : end);
    const uint32_t seq_b = (lseq + 1) >> 1;
    wr32(o, s.size - 4);
    wr32(o + 4, (uint32_t)tid); wr32(o + 8, (uint32_t)pos); wr32(o + 12, (bin << 16) | (255u << 8) | l_rn);
    wr32(o + 16, ((anti ? 0x10u : 0u) << 16) | (uint32_t)a.n_cigar); wr32(o + 20, lseq); wr32(o + 24, 0xFFFFFFFFu); wr32(o + 28, 0xFFFFFFFFu); wr32(o + 32, 0);
    for (uint32_t k = 0; k < l_rn; ++k) o[36 + k] = raw[32 + k];
    uint8_t* oc = o + 36 + l_rn;
    for (int i = 0; i < a.n_cigar; ++i) wr32(oc + 4 * i, ((a.cigar[i] & 0x0FFFFFFFu) << 4) | bam_op(a.cigar[i] >> 28));
    uint8_t* os = oc + 4 * (uint32_t)a.n_cigar;
    uint8_t* oq = os + seq_b;
    if (!anti) {
        for (uint32_t k = 0; k < seq_b; ++k) os[k] = sq[k];
        if (lseq & 1u) os[seq_b - 1] &= 0xF0u;
        for (uint32_t k = 0; k < lseq; ++k) oq[k] = ql[k];
    } else {
        for (uint32_t b = 0; b < seq_b; ++b) {
            uint32_t v = 0;
            for (uint32_t h = 0; h < 2; ++h) {
                const uint32_t k = 2 * b + h;
                if (k >= lseq) break;
                const uint32_t j = lseq - 1 - k;
                const uint32_t nib = (sq[j >> 1] >> ((j & 1u) ? 0 : 4)) & 0xFu;
                const uint32_t comp = nib == 1 ? 8u : nib == 2 ? 4u : nib == 4 ? 2u : nib == 8 ? 1u : 15u;
                v |= comp << (h ? 0 : 4);
            }
            os[b] = (uint8_t)v;
        }
        for (uint32_t k = 0; k < lseq; ++k) oq[k] = ql[lseq - 1 - k];
    }
    uint8_t* t = oq + lseq;
    t = put_int(t, 'A', 'S', (int)a.AS); t = put_int(t, 'X', 'M', (int)a.XM); t = put_int(t, 'X', 'O', (int)a.XO); t = put_int(t, 'X', 'G', (int)a.XG);
    t[0] = 'M'; t[1] = 'D'; t[2] = 'Z';
    for (uint32_t k = 0; k < a.md_len; ++k) t[3 + k] = (uint8_t)a.md[k];
    t[3 + a.md_len] = 0;
    t += 4 + a.md_len;
    t = put_int(t, 'N', 'M', (int)a.mismatches + s.indel);
    if (s.spliced) { t[0] = 'X'; t[1] = 'S'; t[2] = 'A'; t[3] = (a.flags & THJ_HIT_ANTISENSE_SPLICE) ? '-' : '+'; }
}

}  // namespace bamenc
