#include "xgfa.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string_view>
#include <unordered_map>
#include <unordered_set>

namespace {

// Labels of one block: gap-stripped MSA[i].substr(prev, end-prev+1), std::string::substr clamping
struct BlockLabels {
    std::vector<uint8_t> arena;
    std::vector<std::string_view> label;   // one per row, empty = skipped row (fbg.cpp:1215,1235)
    void build(const Msa &msa, uint64_t prev, uint64_t end)
    {
        const uint64_t n = msa.n, m = msa.m;
        const uint64_t stop = std::min(end + 1, n);
        const uint64_t width = stop > prev ? stop - prev : 0;
        arena.resize(m * width + 1);
        label.assign(m, std::string_view());
        for (uint64_t i = 0; i < m; i++) {
            const uint8_t *row = msa.cells.data() + i * n;
            uint8_t *dst = arena.data() + i * width;
            uint64_t k = 0;
            for (uint64_t j = prev; j < stop; j++)
                if (row[j] != '-') dst[k++] = row[j];
            label[i] = std::string_view(reinterpret_cast<const char *>(dst), k);
        }
    }
};

struct Writer {
    FILE *fp;
    std::vector<char> buf;
    size_t used = 0;
    bool ok = true;
    explicit Writer(FILE *f) : fp(f), buf(1 << 22) {}
    void flush() { if (used && std::fwrite(buf.data(), 1, used, fp) != used) ok = false; used = 0; }
    void raw(const char *p, size_t len)
    {
        if (len > buf.size() - used) flush();
        if (len > buf.size()) { if (std::fwrite(p, 1, len, fp) != len) ok = false; return; }
        std::memcpy(buf.data() + used, p, len); used += len;
    }
    void str(const char *s) { raw(s, std::strlen(s)); }
    void num(uint64_t v)
    {
        char t[24]; int k = 24;
        do { t[--k] = char('0' + v % 10); v /= 10; } while (v);
        raw(t + k, 24 - k);
    }
};

} // namespace

bool write_xgfa(const Msa &msa, const std::vector<uint64_t> &boundaries, bool output_paths,
                const std::string &path, std::string &error)
{
    FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp) { error = "cannot open " + path + " for writing"; return false; }
    Writer w(fp);
    const uint64_t m = msa.m, nb = boundaries.size();
    BlockLabels bl;

    w.str("M\t"); w.num(m); w.str("\t"); w.num(msa.n); w.str("\n");                 // fbg.cpp:1201
    w.str("X\t1");                                                                 // 1204-1207
    for (uint64_t i = 0; i + 1 < nb; i++) { w.str("\t"); w.num(boundaries[i] + 2); }
    w.str("\n");

    w.str("B\t");                                                                  // 1210-1220
    {
        std::unordered_set<std::string_view> labels;
        uint64_t prev = 0;
        for (uint64_t j = 0; j < nb; prev = boundaries[j] + 1, j++) {
            bl.build(msa, prev, boundaries[j]);
            labels.clear();
            for (uint64_t i = 0; i < m; i++)
                if (!bl.label[i].empty()) labels.insert(bl.label[i]);
            if (j) w.str("\t");
            w.num(labels.size());
        }
    }
    w.str("\n");

    // nodes and edges, block by block (1224-1260)
    std::vector<std::vector<uint64_t>> paths;
    if (output_paths) paths.assign(m, {});
    {
        std::unordered_map<std::string_view, uint64_t> cur;
        std::vector<int64_t> row_prev(m, -1), row_cur(m, -1);
        std::vector<std::pair<uint64_t, uint64_t>> edges;
        uint64_t nodecount = 0, prev = 0;
        for (uint64_t j = 0; j < nb; prev = boundaries[j] + 1, j++) {
            bl.build(msa, prev, boundaries[j]);
            cur.clear();
            edges.clear();
            for (uint64_t i = 0; i < m; i++) {
                row_cur[i] = -1;
                const std::string_view lab = bl.label[i];
                if (lab.empty()) continue;
                auto it = cur.find(lab);
                uint64_t id;
                if (it == cur.end()) {
                    id = nodecount++;
                    cur.emplace(lab, id);
                    w.str("S\t"); w.num(id); w.str("\t"); w.raw(lab.data(), lab.size()); w.str("\n");   // 1241
                } else {
                    id = it->second;
                }
                row_cur[i] = (int64_t)id;
                if (row_prev[i] >= 0) edges.emplace_back((uint64_t)row_prev[i], id);                    // 1249-1250
                if (output_paths) paths[i].push_back(id);                                               // 1267-1289
            }
            std::sort(edges.begin(), edges.end());                                                      // std::set order
            edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
            for (const auto &e : edges) {                                                               // 1253-1255
                w.str("L\t"); w.num(e.first); w.str("\t+\t"); w.num(e.second); w.str("\t+\t0M\n");
            }
            row_prev.swap(row_cur);
        }
    }
    if (output_paths) {                                                                                 // 1291-1300
        if (msa.identifiers.size() != m) {   // assert(identifiers.size() == paths.size()) in the reference
            error = "number of FASTA headers differs from the number of rows kept (the reference asserts here, "
                    "fbg.cpp:1292)";
            std::fclose(fp);
            return false;
        }
        for (uint64_t i = 0; i < m; i++) {
            if (paths[i].empty()) {          // reference underflows size()-1 (fbg.cpp:1295): undefined
                error = "row " + std::to_string(i) + " has no non-gap character; its P line is undefined in the "
                        "reference (fbg.cpp:1295)";
                std::fclose(fp);
                return false;
            }
            w.str("P\t"); w.raw(msa.identifiers[i].data(), msa.identifiers[i].size()); w.str("\t");
            for (size_t k = 0; k + 1 < paths[i].size(); k++) { w.num(paths[i][k]); w.str("+,"); }
            w.num(paths[i].back()); w.str("+"); w.str("\t*\n");
        }
    }
    w.flush();
    const bool ok = w.ok && std::fclose(fp) == 0;
    if (!ok) error = "write to " + path + " failed";
    return ok;
}

bool write_xgfa_graph(const Msa &msa, const std::vector<uint64_t> &boundaries, const BlockGraph &g, bool output_paths,
                      const std::string &path, std::string &error)
{
    FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp) { error = "cannot open " + path + " for writing"; return false; }
    Writer w(fp);
    const uint64_t m = msa.m, n = msa.n, nb = boundaries.size();
    w.str("M\t"); w.num(m); w.str("\t"); w.num(n); w.str("\n");                     // fbg.cpp:1201
    w.str("X\t1");                                                                 // 1204-1207
    for (uint64_t i = 0; i + 1 < nb; i++) { w.str("\t"); w.num(boundaries[i] + 2); }
    w.str("\n");
    w.str("B\t");                                                                  // 1210-1220
    for (uint64_t j = 0; j < nb; j++) { if (j) w.str("\t"); w.num(g.first_node[j + 1] - g.first_node[j]); }
    w.str("\n");
    std::vector<char> label;
    uint64_t prev = 0;
    for (uint64_t j = 0; j < nb; prev = boundaries[j] + 1, j++) {                  // 1224-1260
        const uint64_t stop = std::min(boundaries[j] + 1, n);
        const uint64_t cnt = g.first_node[j + 1] - g.first_node[j];
        for (uint64_t k = 0; k < cnt; k++) {
            const uint8_t *row = msa.cells.data() + (uint64_t)g.rep_row[j * m + k] * n;
            label.clear();
            for (uint64_t x = prev; x < stop; x++)
                if (row[x] != '-') label.push_back((char)row[x]);
            w.str("S\t"); w.num(g.first_node[j] + k); w.str("\t"); w.raw(label.data(), label.size()); w.str("\n");   // 1241
        }
        for (uint64_t e = 0; e < g.edge_count[j]; e++) {                            // 1253-1255
            const uint64_t pr = g.edges[j * m + e];
            w.str("L\t"); w.num(pr >> 32); w.str("\t+\t"); w.num(pr & 0xffffffffu); w.str("\t+\t0M\n");
        }
    }
    if (output_paths) {                                                                                 // 1291-1300
        if (msa.identifiers.size() != m) {
            error = "number of FASTA headers differs from the number of rows kept (the reference asserts here, "
                    "fbg.cpp:1292)";
            std::fclose(fp);
            return false;
        }
        std::vector<uint32_t> path_ids;
        for (uint64_t i = 0; i < m; i++) {
            path_ids.clear();
            for (uint64_t j = 0; j < nb; j++)
                if (g.node_of[j * m + i] != 0xffffffffu) path_ids.push_back(g.node_of[j * m + i]);
            if (path_ids.empty()) {
                error = "row " + std::to_string(i) + " has no non-gap character; its P line is undefined in the "
                        "reference (fbg.cpp:1295)";
                std::fclose(fp);
                return false;
            }
            w.str("P\t"); w.raw(msa.identifiers[i].data(), msa.identifiers[i].size()); w.str("\t");
            for (size_t k = 0; k + 1 < path_ids.size(); k++) { w.num(path_ids[k]); w.str("+,"); }
            w.num(path_ids.back()); w.str("+"); w.str("\t*\n");
        }
    }
    w.flush();
    const bool ok = w.ok && std::fclose(fp) == 0;
    if (!ok) error = "write to " + path + " failed";
    return ok;
}

GraphStats segment_stats(const Msa &msa, const std::vector<uint64_t> &boundaries)
{
    // fbg.cpp:667-728: labels deduplicated globally; blocks[j] holds only the nodes first seen in block j
    GraphStats st;
    std::unordered_map<std::string, uint64_t> str2id;
    std::vector<uint64_t> prev_ids(msa.m), ids(msa.m);
    std::unordered_set<uint64_t> edges;   // src * 2^32 + dst is not enough in general: use a pair hash via string
    std::unordered_map<uint64_t, std::unordered_set<uint64_t>> adj;
    BlockLabels bl;
    uint64_t prev = 0;
    for (uint64_t j = 0; j < boundaries.size(); j++) {
        bl.build(msa, prev, boundaries[j]);
        uint64_t fresh = 0;
        for (uint64_t i = 0; i < msa.m; i++) {
            std::string lab(bl.label[i]);
            auto it = str2id.find(lab);
            if (it == str2id.end()) {
                st.total_label_length += lab.size();
                it = str2id.emplace(std::move(lab), st.nodes++).first;
                fresh++;
            }
            ids[i] = it->second;
        }
        st.founders = std::max(st.founders, fresh);
        if (j > 0)
            for (uint64_t i = 0; i < msa.m; i++) adj[prev_ids[i]].insert(ids[i]);
        prev_ids.swap(ids);
        prev = boundaries[j] + 1;
    }
    for (const auto &kv : adj) st.edges += kv.second.size();
    return st;
}
