// host_selftest.cpp -- the host-side pieces of the program that need no GPU (FASTA input with the reference's row
// filters, xGFA writer, graph statistics) behind a command line, for the CPU tests (tests/test_host_io.py) and for
// sanitizer runs:
//     fbg_host_selftest FASTA GAP_LIMIT ELASTIC(0|1) PATHS(0|1) OUT.gfa [GRAPH=file] [BOUNDARY ...]
// GRAPH=file: the node / edge arrays as fbg_block_graph returns them (u64 nb, u64 m, u32 node_of[nb*m], u32 rep_row[nb*m],
// u64 first_node[nb+1], u64 edge_count[nb], u64 edges[nb*m]); the xGFA is then formatted from them (write_xgfa_graph),
// as the program does with the GPU's arrays, instead of from labels hashed on the host (write_xgfa).
// prints "m n" of the MSA as read, then (with boundaries given: inclusive block ends, the last one == n, fbg.cpp:2027-2039)
// writes the xGFA and prints "nodes total_label_length founders edges".
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "fasta.hpp"
#include "xgfa.hpp"

int main(int argc, char **argv)
{
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s FASTA GAP_LIMIT ELASTIC PATHS OUT.gfa [BOUNDARY ...]\n", argv[0]);
        return 2;
    }
    Msa msa;
    if (!read_msa(argv[1], std::atol(argv[2]), std::atoi(argv[3]) != 0, std::atoi(argv[4]) != 0, msa)) {
        std::fprintf(stderr, "cannot read %s\n", argv[1]);
        return 3;
    }
    std::printf("%llu %llu\n", (unsigned long long)msa.m, (unsigned long long)msa.n);
    if (argc == 6 || msa.m == 0) return 0;
    std::vector<uint64_t> boundaries;
    int first_boundary = 6;
    BlockGraph g;
    bool have_graph = false;
    if (std::string(argv[6]).rfind("GRAPH=", 0) == 0) {
        first_boundary = 7;
        have_graph = true;
        FILE *fh = std::fopen(argv[6] + 6, "rb");
        uint64_t hdr[2];
        if (!fh || std::fread(hdr, 8, 2, fh) != 2) { std::fprintf(stderr, "cannot read %s\n", argv[6] + 6); return 5; }
        const uint64_t nb = hdr[0], m = hdr[1];
        g.node_of.resize(nb * m); g.rep_row.resize(nb * m); g.first_node.resize(nb + 1); g.edge_count.resize(nb); g.edges.resize(nb * m);
        const bool ok = std::fread(g.node_of.data(), 4, nb * m, fh) == nb * m && std::fread(g.rep_row.data(), 4, nb * m, fh) == nb * m &&
                        std::fread(g.first_node.data(), 8, nb + 1, fh) == nb + 1 && std::fread(g.edge_count.data(), 8, nb, fh) == nb &&
                        std::fread(g.edges.data(), 8, nb * m, fh) == nb * m;
        std::fclose(fh);
        if (!ok || m != msa.m) { std::fprintf(stderr, "%s does not hold the arrays of this MSA\n", argv[6] + 6); return 5; }
    }
    for (int i = first_boundary; i < argc; i++) boundaries.push_back(std::strtoull(argv[i], nullptr, 10));
    std::string error;
    const bool paths = std::atoi(argv[4]) != 0;
    if (!(have_graph ? write_xgfa_graph(msa, boundaries, g, paths, argv[5], error) : write_xgfa(msa, boundaries, paths, argv[5], error))) {
        std::fprintf(stderr, "%s\n", error.c_str());
        return 4;
    }
    const GraphStats st = segment_stats(msa, boundaries);
    std::printf("%llu %llu %llu %llu\n", (unsigned long long)st.nodes, (unsigned long long)st.total_label_length,
                (unsigned long long)st.founders, (unsigned long long)st.edges);
    return 0;
}
