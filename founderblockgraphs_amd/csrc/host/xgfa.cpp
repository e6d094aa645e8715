#include "xgfa.hpp"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <condition_variable>
#include <mutex>
#include <thread>

static unsigned g_host_threads = 0;      // 0 = as many as the machine has, at most 16
void set_host_threads(unsigned t) { g_host_threads = t; }

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

namespace {

// append-only text buffer with the same number / string helpers as Writer
struct Chunk {
    std::vector<char> d;
    void raw(const char *p, size_t len) { d.insert(d.end(), p, p + len); }
    void str(const char *s) { raw(s, std::strlen(s)); }
    void num(uint64_t v)
    {
        char t[24]; int k = 24;
        do { t[--k] = char('0' + v % 10); v /= 10; } while (v);
        raw(t + k, 24 - k);
    }
};

// Runs make(unit, chunk) for unit = 0 .. units-1 on a few threads and writes the chunks to fp in unit order.
// At most `window` finished-but-unwritten units are kept, so memory stays bounded whatever the output size.
template <class Make> bool ordered_parallel_write(FILE *fp, uint64_t units, Make make)
{
    if (units == 0) return true;
    unsigned T = std::thread::hardware_concurrency();
    if (T == 0) T = 1;
    if (T > 16) T = 16;
    if (g_host_threads && g_host_threads < T) T = g_host_threads;
    if ((uint64_t)T > units) T = (unsigned)units;
    const uint64_t window = 4 * (uint64_t)T;
    std::vector<Chunk> slot(window);
    std::vector<char> ready(window, 0);
    std::mutex mu;
    std::condition_variable cv_ready, cv_space;
    uint64_t next_unit = 0, written = 0;
    bool ok = true;
    auto worker = [&]() {
        for (;;) {
            uint64_t u;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_space.wait(lk, [&] { return next_unit >= units || next_unit < written + window; });
                if (next_unit >= units) return;
                u = next_unit++;
            }
            Chunk &c = slot[u % window];
            c.d.clear();
            make(u, c);
            {
                std::lock_guard<std::mutex> lk(mu);
                ready[u % window] = 1;
            }
            cv_ready.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; t++) pool.emplace_back(worker);
    for (uint64_t u = 0; u < units; u++) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_ready.wait(lk, [&] { return ready[u % window] != 0; });
        }
        Chunk &c = slot[u % window];
        if (!c.d.empty() && std::fwrite(c.d.data(), 1, c.d.size(), fp) != c.d.size()) ok = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            ready[u % window] = 0;
            written = u + 1;
        }
        cv_space.notify_all();
    }
    for (auto &th : pool) th.join();
    return ok;
}

} // namespace

bool write_xgfa_graph(const Msa &msa, const std::vector<uint64_t> &boundaries, const BlockGraph &g, bool output_paths,
                      const std::string &path, std::string &error)
{
    FILE *fp = std::fopen(path.c_str(), "wb");
    if (!fp) { error = "cannot open " + path + " for writing"; return false; }
    std::setvbuf(fp, nullptr, _IOFBF, 1 << 22);
    const uint64_t m = msa.m, n = msa.n, nb = boundaries.size();
    bool ok = true;
    {
        Writer w(fp);
        w.str("M\t"); w.num(m); w.str("\t"); w.num(n); w.str("\n");                     // fbg.cpp:1201
        w.str("X\t1");                                                                 // 1204-1207
        for (uint64_t i = 0; i + 1 < nb; i++) { w.str("\t"); w.num(boundaries[i] + 2); }
        w.str("\n");
        w.str("B\t");                                                                  // 1210-1220
        for (uint64_t j = 0; j < nb; j++) { if (j) w.str("\t"); w.num(g.first_node[j + 1] - g.first_node[j]); }
        w.str("\n");
        w.flush();
        ok = w.ok;
    }
    // S and L lines, block by block (1224-1260): formatted by several threads, written in block order
    const uint64_t per_unit = 256;
    ok = ordered_parallel_write(fp, (nb + per_unit - 1) / per_unit, [&](uint64_t u, Chunk &c) {
        const uint64_t j0 = u * per_unit, j1 = std::min(nb, j0 + per_unit);
        std::vector<char> label;
        for (uint64_t j = j0; j < j1; j++) {
            const uint64_t prev = j ? boundaries[j - 1] + 1 : 0;
            const uint64_t stop = std::min(boundaries[j] + 1, n);
            const uint64_t cnt = g.first_node[j + 1] - g.first_node[j];
            for (uint64_t k = 0; k < cnt; k++) {
                const uint8_t *row = msa.cells.data() + (uint64_t)g.rep_row[j * m + k] * n;
                label.clear();
                for (uint64_t x = prev; x < stop; x++)
                    if (row[x] != '-') label.push_back((char)row[x]);
                c.str("S\t"); c.num(g.first_node[j] + k); c.str("\t"); c.raw(label.data(), label.size()); c.str("\n");   // 1241
            }
            for (uint64_t e = 0; e < g.edge_count[j]; e++) {                            // 1253-1255
                const uint64_t pr = g.edges[j * m + e];
                c.str("L\t"); c.num(pr >> 32); c.str("\t+\t"); c.num(pr & 0xffffffffu); c.str("\t+\t0M\n");
            }
        }
    }) && ok;
    if (output_paths) {                                                                                 // 1291-1300
        if (msa.identifiers.size() != m) {
            error = "number of FASTA headers differs from the number of rows kept (the reference asserts here, "
                    "fbg.cpp:1292)";
            std::fclose(fp);
            return false;
        }
        for (uint64_t i = 0; i < m; i++) {
            bool any = false;
            for (uint64_t j = 0; j < nb && !any; j++) any = g.node_of[j * m + i] != 0xffffffffu;
            if (!any) {
                error = "row " + std::to_string(i) + " has no non-gap character; its P line is undefined in the "
                        "reference (fbg.cpp:1295)";
                std::fclose(fp);
                return false;
            }
        }
        ok = ordered_parallel_write(fp, m, [&](uint64_t i, Chunk &c) {
            c.str("P\t"); c.raw(msa.identifiers[i].data(), msa.identifiers[i].size()); c.str("\t");
            bool first = true;
            for (uint64_t j = 0; j < nb; j++) {
                const uint32_t id = g.node_of[j * m + i];
                if (id == 0xffffffffu) continue;
                if (!first) c.str("+,");
                c.num(id);
                first = false;
            }
            c.str("+"); c.str("\t*\n");
        }) && ok;
    }
    ok = (std::fclose(fp) == 0) && ok;
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
