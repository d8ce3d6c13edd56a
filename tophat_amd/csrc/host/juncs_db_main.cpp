// juncs_db -- MI355X-native drop-in for TopHat's juncs_db (same argv, FASTA on stdout; tophat.py:2574-2586, parsed like
// juncs_db.cpp:62-296).  The sequences around every junction / deletion / insertion / fusion are gathered from the
// bit-plane genome resident on the device (thj_genome_gather); this host side reads the coordinate lists into the
// reference's std::set orders, decides which records print_splice / print_insertion / print_fusion would emit
// (juncs_db.cpp:73-233) and writes their header lines around the gathered bases.
#include "thj_hostio.h"

#include <set>

using namespace thjh;

static void print_usage() {
    fprintf(stderr, "Usage:   juncs_db <min_anchor> <read_length> <splice_coords1,...,splice_coordsN> <insertion_coords1,...,insertion_coordsN> "
                    "<deletion_coords1,...,deletion_coordsN> <fusion_coords1,...,fusion_coordsN> <ref.fa>\n");
}

struct J { uint32_t ref, left, right; bool anti; };
struct JLess {                                             // junctions.h:39-57
    bool operator()(const J& a, const J& b) const {
        if (a.ref != b.ref) return a.ref < b.ref;
        if (a.left != b.left) return a.left < b.left;
        if (a.right != b.right) return a.right < b.right;
        return a.anti < b.anti;
    }
};
struct I { uint32_t ref, left; std::string seq; };
struct ILess {                                             // insertions.h:52-67: the sequence only by its length
    bool operator()(const I& a, const I& b) const {
        if (a.ref != b.ref) return a.ref < b.ref;
        if (a.left != b.left) return a.left < b.left;
        return a.seq.size() < b.seq.size();
    }
};
struct F { uint32_t r1, r2, left, right, dir; };
struct FLess {                                             // fusions.h:38-69
    bool operator()(const F& a, const F& b) const {
        if (a.r1 != b.r1) return a.r1 < b.r1;
        if (a.r2 != b.r2) return a.r2 < b.r2;
        if (a.left != b.left) return a.left < b.left;
        if (a.right != b.right) return a.right < b.right;
        return a.dir < b.dir;
    }
};

// get_token / strsep over tab-separated fields: nullptr when the line has run out
static char* next_field(char** buf) {
    if (!*buf) return nullptr;
    char* s = *buf;
    char* t = strchr(s, '\t');
    if (t) { *t = 0; *buf = t + 1; } else *buf = nullptr;
    return s;
}

static std::vector<FILE*> open_list(const std::string& list) {
    std::vector<FILE*> out;
    for (auto& fn : split(list, ',')) {
        FILE* f = fopen(fn.c_str(), "r");
        if (!f) { fprintf(stderr, "Warning: cannot open %s for reading\n", fn.c_str()); continue; }
        out.push_back(f);
    }
    return out;
}

int main(int argc, char** argv) {
    fprintf(stderr, "juncs_db (MI355X-native, %s)\n---------------------------\n", thj_version());
    Opts o;
    int rc = parse_options(argc, argv, o, print_usage);
    if (rc) return rc;
    std::vector<std::string> pos;
    for (int i = optind; i < argc; ++i) pos.push_back(argv[i]);
    if (pos.size() < 7) { print_usage(); return 1; }
    const int min_anchor_len = atoi(pos[0].c_str());
    if (min_anchor_len < 3) { fprintf(stderr, "anchor length must be at least 3\n"); print_usage(); return 1; }
    const int read_length = atoi(pos[1].c_str());
    if (read_length < 4) { fprintf(stderr, "read length must be at least 4\n"); print_usage(); return 1; }
    std::vector<FILE*> jf = open_list(pos[2]), inf = open_list(pos[3]), df = open_list(pos[4]), ff = open_list(pos[5]);

    RefTable rt;
    rt.load_sam_header(o.sam_header);
    rt.load_fasta(pos[6]);

    char line[2048];
    auto chomp = [&]() { char* nl = strrchr(line, '\n'); if (nl) *nl = 0; };
    std::set<J, JLess> juncs, dels;
    for (FILE* f : jf) {
        while (fgets(line, sizeof line, f)) {
            chomp();
            char* buf = line;
            char* name = next_field(&buf); char* l = next_field(&buf); char* r = next_field(&buf); char* ori = next_field(&buf);
            if (!l || !r || !ori) { fprintf(stderr, "Error: malformed splice coordinate record\n"); return 1; }
            juncs.insert({rt.get_id(name), (uint32_t)atoi(l), (uint32_t)atoi(r), *ori == '-'});
        }
        fclose(f);
    }
    for (FILE* f : df) {
        while (fgets(line, sizeof line, f)) {
            chomp();
            char* buf = line;
            char* name = next_field(&buf); char* l = next_field(&buf); char* r = next_field(&buf);
            if (!l || !r) { fprintf(stderr, "Error: malformed deletion coordinate record\n"); return 1; }
            dels.insert({rt.get_id(name), (uint32_t)atoi(l) - 1u, (uint32_t)atoi(r), false});      // :385
        }
        fclose(f);
    }
    std::set<I, ILess> ins;
    for (FILE* f : inf) {
        while (fgets(line, sizeof line, f)) {
            chomp();
            char* buf = line;
            char* name = next_field(&buf); char* l = next_field(&buf); char* r = next_field(&buf); char* sq = next_field(&buf);
            if (!l || !sq || !r) { fprintf(stderr, "Error: malformed insertion coordinate record\n"); return 1; }
            std::string seq;                                     // Dna5String(scan_sequence): anything else becomes N
            bool has_n = false;
            for (const char* c = sq; *c; ++c) {
                char u = (char)toupper((unsigned char)*c);
                if (u != 'A' && u != 'C' && u != 'G' && u != 'T') { has_n = true; break; }
                seq.push_back(u);
            }
            if (has_n) continue;                                 // no ambiguities in the insertion (:416-428)
            ins.insert({rt.get_id(name), (uint32_t)atoi(l), seq});
        }
        fclose(f);
    }
    std::set<F, FLess> fus;
    for (FILE* f : ff) {
        while (fgets(line, sizeof line, f)) {
            chomp();
            char* buf = line;
            char* n1 = next_field(&buf); char* l = next_field(&buf); char* n2 = next_field(&buf); char* r = next_field(&buf); char* d = next_field(&buf);
            if (!n1 || !l || !n2 || !r || !d) { fprintf(stderr, "Error: malformed insertion coordinate record\n"); return 1; }
            uint32_t dir = THJ_FUSION_FF;
            if (!strcmp(d, "fr")) dir = THJ_FUSION_FR; else if (!strcmp(d, "rf")) dir = THJ_FUSION_RF; else if (!strcmp(d, "rr")) dir = THJ_FUSION_RR;
            uint32_t id1 = rt.get_id(n1), id2 = rt.get_id(n2);
            fus.insert({id1, id2, (uint32_t)atoi(l), (uint32_t)atoi(r), dir});
        }
        fclose(f);
    }

    // ---- text layout: header lines written here, the bases left for the device
    std::string text;
    std::vector<thj_piece> pieces;
    std::vector<int64_t> off;
    auto piece = [&](uint32_t ref, size_t start, size_t end, bool rcomp) {
        pieces.push_back({ref, (int32_t)start, (int32_t)(end - start), rcomp ? THJ_PIECE_RC : 0u});
        off.push_back((int64_t)text.size());
        text.append(end - start, '?');
    };
    auto has_seq = [&](uint32_t ref) { return ref >= 1 && ref <= rt.seqs.size() && !rt.seqs[ref - 1].empty(); };
    auto splice = [&](const J& j, const char* tag) {                            // print_splice :111-164
        if (!has_seq(j.ref)) return;
        const size_t ref_len = rt.seqs[j.ref - 1].size();
        const int half = read_length;
        if (!(j.left <= ref_len && j.right <= ref_len)) return;
        size_t left_start = (int)j.left - half + 1 >= 0 ? (size_t)((int)j.left - half + 1) : 0;
        size_t left_end = left_start + (size_t)half;
        size_t right_start = j.right;
        size_t right_end = right_start + (size_t)half < ref_len ? right_start + (size_t)half : ref_len;
        if (!(left_start < left_end && left_end <= ref_len && right_start < right_end && right_end <= ref_len)) return;
        text += ">" + rt.names[j.ref - 1] + "|" + std::to_string(left_start) + "|" + std::to_string(j.left) + "-" + std::to_string(j.right) + "|" +
                std::to_string(right_end) + "|" + tag + "\n";
        piece(j.ref, left_start, left_end, false);
        piece(j.ref, right_start, right_end, false);
        text += "\n";
    };
    for (auto& j : juncs) splice(j, j.anti ? "GTAG|rev" : "GTAG|fwd");
    for (auto& j : dels) splice(j, j.anti ? "del|rev" : "del|fwd");
    for (auto& x : ins) {                                                       // print_insertion :73-108
        if (!has_seq(x.ref)) continue;
        const size_t ref_len = rt.seqs[x.ref - 1].size();
        const int half = read_length - min_anchor_len;
        if (!(x.left <= ref_len)) continue;
        size_t left_start = (int)x.left - half + 1 >= 0 ? (size_t)((int)x.left - half + 1) : 0;
        size_t left_end = left_start + (size_t)half;
        size_t right_start = (size_t)(int)left_end;
        size_t right_end = right_start + (size_t)half < ref_len ? right_start + (size_t)half : ref_len;
        if (!(left_start < left_end && left_end <= ref_len && right_start < right_end && right_end <= ref_len)) continue;
        text += ">" + rt.names[x.ref - 1] + "|" + std::to_string(left_start) + "|" + std::to_string(x.left) + "-" + x.seq + "|" +
                std::to_string(right_end) + "|ins|fwd\n";
        piece(x.ref, left_start, left_end, false);
        text += x.seq;
        piece(x.ref, right_start, right_end, false);
        text += "\n";
    }
    for (auto& x : fus) {                                                       // print_fusion :166-233
        if (!has_seq(x.r1) || !has_seq(x.r2)) continue;
        const size_t llen = rt.seqs[x.r1 - 1].size(), rlen = rt.seqs[x.r2 - 1].size();
        const int half = read_length - min_anchor_len;
        if (!(x.left < llen && x.right < rlen)) continue;
        size_t left_start, left_end, right_start, right_end;
        if (x.dir == THJ_FUSION_FF || x.dir == THJ_FUSION_FR) {
            left_start = (size_t)x.left + 1 >= (size_t)half ? (size_t)(x.left - (uint32_t)half + 1u) : 0;
            left_end = left_start + (size_t)half;
        } else {
            left_start = x.left;
            left_end = left_start + (size_t)half < llen ? left_start + (size_t)half : llen;
        }
        if (x.dir == THJ_FUSION_FF || x.dir == THJ_FUSION_RF) {
            right_start = x.right;
            right_end = right_start + (size_t)half < rlen ? right_start + (size_t)half : rlen;
        } else {
            right_end = (size_t)x.right + 1;
            right_start = right_end >= (size_t)half ? right_end - (size_t)half : 0;
        }
        if (!(left_start < left_end && left_end <= llen && right_start < right_end && right_end <= rlen)) continue;
        const bool lrc = x.dir == THJ_FUSION_RF || x.dir == THJ_FUSION_RR, rrc = x.dir == THJ_FUSION_FR || x.dir == THJ_FUSION_RR;
        const size_t ls_print = lrc ? left_end - 1 : left_start, re_print = rrc ? right_start - 1 : right_end;
        const char* d = x.dir == THJ_FUSION_FR ? "fr" : x.dir == THJ_FUSION_RF ? "rf" : x.dir == THJ_FUSION_RR ? "rr" : "ff";
        text += ">" + rt.names[x.r1 - 1] + "-" + rt.names[x.r2 - 1] + "|" + std::to_string(ls_print) + "|" + std::to_string(x.left) + "-" +
                std::to_string(x.right) + "|" + std::to_string(re_print) + "|fus|" + d + "\n";
        piece(x.r1, left_start, left_end, lrc);
        piece(x.r2, right_start, right_end, rrc);
        text += "\n";
    }

    if (!pieces.empty()) {
        int device = getenv("THJ_DEVICE") ? atoi(getenv("THJ_DEVICE")) : 0;
        thj_ctx* ctx = nullptr;
        if (thj_ctx_create(device, nullptr, &ctx)) die("Error: %s\n", thj_last_error());
        rt.upload(ctx);
        if (thj_genome_gather(ctx, pieces.data(), (int64_t)pieces.size(), off.data(), &text[0], (int64_t)text.size())) die("Error: %s\n", thj_last_error());
        thj_ctx_destroy(ctx);
    }
    fwrite(text.data(), 1, text.size(), stdout);
    return 0;
}
