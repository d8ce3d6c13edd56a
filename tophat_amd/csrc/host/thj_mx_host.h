// thj_mx_host.h -- the host's part of the microexon search: add_to_microexon_windows (segment_juncs.cpp:3675-3735) over the candidate
// windows the device found (thj_microexon_collect / thj_microexon_candidates), in the order the reference visits them: all reads of the
// left side, then all reads of the right side, each read's hits in list order.  Sequential std::map logic, as in the reference (which
// runs this mode on one thread, :4719-4721); the output feeds thj_microexon_run.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <tuple>
#include <vector>

#include "../../../include/thj.h"

namespace thjh {

struct MxWindows {
    std::vector<thj_mx_window> windows;            // in map order (ref_id, left, right): the order align_microexon_segs walks them
    std::vector<uint64_t> strs; std::vector<uint8_t> str_len; std::vector<uint32_t> str_window;
};

inline bool mx_overlap_in_genome(int ll, int lr, int rl, int rr) {               // :3662-3673
    if (ll >= rl && ll < rr) return true;
    if (lr > rl && lr < rr) return true;
    if (rl >= ll && rl < lr) return true;
    if (rr > ll && rr < lr) return true;
    return false;
}

// cands: any order; sorted here by (side, ordinal, rank) = visiting order (left reads have the lower ordinals of a pair's two sides only
// by convention of the caller, so the side is compared first)
inline MxWindows mx_merge_windows(std::vector<thj_mx_cand> cands) {
    std::sort(cands.begin(), cands.end(), [](const thj_mx_cand& a, const thj_mx_cand& b) { return std::make_tuple(a.side, a.ordinal, a.rank) < std::make_tuple(b.side, b.ordinal, b.rank); });
    struct Key { uint32_t ref; int32_t left, right; bool operator<(const Key& o) const { return std::tie(ref, left, right) < std::tie(o.ref, o.left, o.right); } };   // RefSeg::operator<
    struct Val { int side; std::vector<uint32_t> strs; };                          // indices into cands
    std::map<Key, Val> m;
    for (uint32_t ci = 0; ci < (uint32_t)cands.size(); ++ci) {
        const thj_mx_cand& c = cands[ci];
        const int left_boundary = c.left, right_boundary = c.right;
        Key left_dummy{c.ref_id, left_boundary, right_boundary};
        const Key right_dummy{c.ref_id, right_boundary, right_boundary + 1};
        auto lb = m.lower_bound(left_dummy);
        const auto ub = m.lower_bound(right_dummy);
        if (lb == m.end()) { m.insert({left_dummy, Val{c.side, {ci}}}); continue; }
        auto first_to_be_erased = m.end();
        auto last_to_be_erased = ub;
        bool have_new = false;
        std::vector<uint32_t> new_vec;
        for (; lb != ub; ++lb) {
            if (mx_overlap_in_genome(lb->first.left, lb->first.right, left_boundary, right_boundary)) {
                have_new = true;
                if (first_to_be_erased == m.end()) first_to_be_erased = lb;
                left_dummy.left = std::min(lb->first.left, left_boundary);
                left_dummy.right = std::max(lb->first.right, right_boundary);
                new_vec.insert(new_vec.end(), lb->second.strs.begin(), lb->second.strs.end());
            } else if (first_to_be_erased != m.end()) last_to_be_erased = lb;
        }
        if (first_to_be_erased != m.end()) m.erase(first_to_be_erased, last_to_be_erased);
        if (have_new) new_vec.push_back(ci); else new_vec.assign(1, ci);
        m.insert({left_dummy, Val{c.side, std::move(new_vec)}});                 // map::insert: a key that is there already stays as it is
    }
    MxWindows out;
    for (auto& kv : m) {
        const uint32_t w = (uint32_t)out.windows.size();
        out.windows.push_back(thj_mx_window{kv.first.ref, kv.first.left, kv.first.right, kv.second.side});
        for (uint32_t ci : kv.second.strs) { out.strs.push_back(cands[ci].str); out.str_len.push_back(cands[ci].len); out.str_window.push_back(w); }
    }
    return out;
}

}  // namespace thjh
