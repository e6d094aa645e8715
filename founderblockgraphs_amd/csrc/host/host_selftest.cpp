// host_selftest.cpp -- the host-side pieces of the program that need no GPU (FASTA input with the reference's row
// filters, xGFA writer, graph statistics) behind a command line, for the CPU tests (tests/test_host_io.py) and for
// sanitizer runs:
//     fbg_host_selftest FASTA GAP_LIMIT ELASTIC(0|1) PATHS(0|1) OUT.gfa [BOUNDARY ...]
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
    for (int i = 6; i < argc; i++) boundaries.push_back(std::strtoull(argv[i], nullptr, 10));
    std::string error;
    if (!write_xgfa(msa, boundaries, std::atoi(argv[4]) != 0, argv[5], error)) {
        std::fprintf(stderr, "%s\n", error.c_str());
        return 4;
    }
    const GraphStats st = segment_stats(msa, boundaries);
    std::printf("%llu %llu %llu %llu\n", (unsigned long long)st.nodes, (unsigned long long)st.total_label_length,
                (unsigned long long)st.founders, (unsigned long long)st.edges);
    return 0;
}
